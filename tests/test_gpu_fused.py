"""The fused sparse levels (itd_set_fuse_mode; pyitd_amd/csrc/itd_knotfirst.hpp) on the GPU: whatever they deliver is the
reference's result bit for bit (rows, baselines, knot counts, stop reasons, against the pinned oracle), and what they cannot
deliver — smooth, quantised, NaN-producing input — they report, after which the engine repeats the call level by level."""
import numpy as np
import pytest

from helpers import ROUND4_WRONG_ROWS, assert_bits_equal, chirp, coarse, fuzz_signal, kf_rates_draws, load_golden, sines_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    return cpu_oracle


def _run(P, torch, x, m, mode, L0=3, bases=True, cap=None):
    n = len(x)
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(mode)
    eng.set_fuse_level(L0)
    if cap is not None:
        eng.set_fuse_cap(cap)                # (a suite run under PYITD_FUSE_CAP caps every other engine's fused levels)
    eng.set_fuse_min_samples(65536)          # (the automatic mode fuses from 2 * 2^20 samples per launch sequence by default)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    bs = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda") if bases else None
    torch.cuda.synchronize()
    eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), bs.data_ptr() if bases else None, None)
    s = eng.summary(1)
    out = {"rows": rows[: int(s["n_rows"][0])].cpu().numpy(), "bases": bs[: int(s["n_baselines"][0])].cpu().numpy() if bases else None,
           "stop": ("natural", "timeout")[int(s["stop"][0])], "knots": [int(v) for v in s["knot_counts"][0] if v >= 0],
           "repeats": eng.fuse_repeats}
    eng.close()
    return out


def _check(got, ref, what):
    assert got["stop"] == ref["stop"], what
    assert_bits_equal(got["rows"], ref["rows"], what + " rows")
    if got["bases"] is not None:
        assert_bits_equal(got["bases"], ref["baselines"], what + " baselines")
    assert got["knots"][1: 1 + len(ref["knot_counts"])] == ref["knot_counts"].tolist(), what + " knots per level"


# (name, signal, max_iteration, first fused level, must the fused form deliver it?)  Deep decompositions (>= ~10 levels) meet
# plateaus born from rounding at their deepest levels now and then (differences shrink geometrically from level to level): there the
# fused form may refuse — tests/test_oracle_knotfirst.py's CPU model refuses the very same cases — and the automatic mode repeats.
CASES = [
    ("sines 2^20 f32", lambda: sines_noise(1 << 20), 7, 3, True),
    ("sines 2^20 f32, fused from level 2", lambda: sines_noise(1 << 20, seed=1), 7, 2, True),
    ("sines 2^18 f32, fused from level 5", lambda: sines_noise(1 << 18, seed=2), 7, 5, True),
    ("sines 2^19 f64", lambda: sines_noise(1 << 19, seed=3, dtype=np.float64), 7, 3, True),
    ("ragged length", lambda: sines_noise((1 << 19) + 777, seed=4), 7, 3, True),
    ("12 rows", lambda: sines_noise(1 << 18, seed=5), 10, 3, False),
    ("natural stop inside the fused levels", lambda: sines_noise(1 << 17, seed=6), 20, 3, False),
    ("white noise f64", lambda: fuzz_signal(np.random.default_rng(7), 0, 300000), 9, 3, False),
    ("random walk f32", lambda: fuzz_signal(np.random.default_rng(8), 1, 300000).astype(np.float32), 9, 2, False),
    ("alternating (every sample a knot at level 0)", lambda: fuzz_signal(np.random.default_rng(9), 6, 200000), 9, 3, False),
    ("max_iteration = first fused level", lambda: sines_noise(1 << 17, seed=10), 3, 3, True),
]


@pytest.mark.parametrize("name,make,m,L0,must_fuse", CASES)
def test_fused_levels_deliver_the_oracle_bit_for_bit(P, torch, oracle, name, make, m, L0, must_fuse):
    from pyitd_amd import ITDError
    from pyitd_amd.engine import FUSE_AUTO, FUSE_ONLY
    x = make()
    ref = oracle.itd(x, m)
    _check(_run(P, torch, x, m, FUSE_AUTO, L0), ref, name + " (automatic)")
    try:
        got = _run(P, torch, x, m, FUSE_ONLY, L0)        # ONLY: a refusal is an error, not a silent repeat
    except ITDError:
        assert not must_fuse, name + ": the fused form refused"
        return
    _check(got, ref, name + " (fused only)")             # what it delivers is the reference's result


def test_full_size_headline_signal(P, torch, oracle):
    """BASELINE configs[1] (2^24 samples, 8 levels) through the fused levels, never repeated: rows bit-exact."""
    from pyitd_amd.engine import FUSE_ONLY
    x = sines_noise(1 << 24)
    ref = oracle.itd_lean(x, 7)
    got = _run(P, torch, x, 7, FUSE_ONLY, 3, bases=False)
    assert got["stop"] == ref["stop"] == "timeout"
    assert_bits_equal(got["rows"], ref["rows"], "2^24 rows")
    assert got["knots"][: len(ref["knot_counts"])] == ref["knot_counts"].tolist()      # itd_lean: knots of every level's input


def test_what_the_fused_levels_cannot_deliver_is_reported_and_repeated(P, torch, oracle):
    """Tiled, coarsely quantised and plateau-led (NaN-producing) input: ONLY refuses, AUTO repeats level by level — the result is the
    oracle's either way — and then starts the next calls level by level.  (A float32 chirp — near ties at its extrema — was refused
    up to round 3 and is delivered since the near ties' samples are candidates.)"""
    from pyitd_amd import ITDError
    from pyitd_amd.engine import FUSE_AUTO, FUSE_ONLY
    radio = load_golden("radio8000_input")["x"]
    lead = np.concatenate([np.zeros(3000), sines_noise(1 << 17, seed=11)[3000:].astype(np.float64)])
    refused = 0
    for name, x, m in (("chirp", chirp(1 << 17), 5), ("tiled clip", np.resize(radio, 1 << 18).astype(np.float32), 9),
                       ("quantised", np.round(fuzz_signal(np.random.default_rng(12), 0, 200000) * 3) / 4.0, 9),
                       ("leading plateau", lead, 6)):
        ref = oracle.itd(x, m)
        got = _run(P, torch, x, m, FUSE_AUTO, 3, cap=-1)     # (uncapped: with a cap below its collapse the tiled clip is delivered)
        _check(got, ref, name + " (automatic)")
        try:
            got2 = _run(P, torch, x, m, FUSE_ONLY, 3, cap=-1)
            _check(got2, ref, name + " (fused only)")     # if it did not refuse it must be right
        except ITDError:
            refused += 1
            assert got["repeats"] == 1, name
    assert refused >= 3


def test_few_knots_stop_before_the_fused_levels(P, torch, oracle):
    from pyitd_amd.engine import FUSE_AUTO
    n = 1 << 17
    t = np.arange(n) / n
    for x in (np.sin(2 * np.pi * 3 * t) + 0.3 * t * t, np.linspace(0, 1, n) ** 2, np.sin(2 * np.pi * 40 * t)):
        ref = oracle.itd(x, 7)
        got = _run(P, torch, x, 7, FUSE_AUTO, 3)
        _check(got, ref, "smooth signal")


def test_batches_and_the_drop_in_class(P, torch, oracle):
    """A batch of 2^17-sample signals (the fused levels run per chunk of signals); one signal of the batch is coarsely quantised:
    just that signal is run again level by level (one in nine) and every signal still equals the oracle.  The drop-in class takes
    the same path."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m, B = 1 << 17, 6, 9
    xs = np.stack([sines_noise(n, seed=20 + b, fscale=1 + b / 50.0) for b in range(B)])
    for with_chirp in (False, True):
        x = xs.copy()
        if with_chirp:
            x[4] = coarse(xs[4])
        eng = P.Engine(n, B, 0)
        eng.set_fuse_mode(FUSE_AUTO)
        eng.set_fuse_min_samples(65536)
        xd = torch.from_numpy(x).cuda()
        rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        eng.decompose_dev(xd.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
        s = eng.summary(B)
        assert (eng.fuse_repeats, eng.fuse_signal_repairs) == (0, 1 if with_chirp else 0)
        for b in range(B):
            ref = oracle.itd_lean(x[b], m)
            assert int(s["n_rows"][b]) == ref["rows"].shape[0]
            assert_bits_equal(rows[b, : ref["rows"].shape[0]].cpu().numpy(), ref["rows"], "signal %d" % b)
        eng.close()
    d = P.ITD()
    x = sines_noise(1 << 18, seed=31)
    ref = oracle.itd(x, 7)
    assert_bits_equal(d.itd(x, 7), ref["rows"], "ITD().itd")
    assert_bits_equal(d.get_baselines(), ref["baselines"], "get_baselines")


def test_first_fused_level_is_two_at_least(P):
    """Level 1's launch completes the signal's own knot count and a level-1 list does not fit the workspace: 2 .. max."""
    from pyitd_amd import ITDError
    eng = P.Engine(1 << 16, 1, 0)
    for bad in (-1, 1, 21):
        with pytest.raises(ITDError):
            eng.set_fuse_level(bad)
    eng.set_fuse_level(0)                         # automatic (the default)
    eng.set_fuse_level(2)
    eng.set_fuse_level(20)
    for bad in (1, 8, 48, 128, -16):              # tiles per knot-side workgroup: 0 (automatic), 16, 32 or 64
        with pytest.raises(ITDError):
            eng.set_fuse_range(bad)
    for ok in (16, 32, 64, 0):
        eng.set_fuse_range(ok)
    eng.close()


def test_a_few_refusing_signals_of_a_batch_are_rerun_on_their_own(P, torch, oracle):
    """Signals 40, 51 and 55 of the bench's batch recipe (2^20 samples, 8 levels) grow a knot from rounding at a deep level — two
    neighbours 2^-27 apart at level 3 are an exact tie at level 8 —, which the fused levels refused up to round 3; with both samples
    of every near tie kept as candidates they are delivered.  Two members are coarsely quantised here: itd_get_summary re-runs just
    those level by level, the rest of the batch keeps its fused result, the engine stays in the fused form, and a second summary of
    the same call does no work."""
    n, M = 1 << 20, 7
    ids = list(range(40, 56))
    xs = np.stack([sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0) for b in ids])
    odd = (3, 12)
    for j in odd:
        xs[j] = coarse(xs[j])
    from pyitd_amd.engine import FUSE_AUTO
    eng = P.Engine(n, len(ids), 0)
    eng.set_fuse_mode(FUSE_AUTO)          # (whatever PYITD_FUSE_MODE says: this test is about the automatic mode's repairs)
    x = torch.from_numpy(xs).cuda()
    rows = torch.full((len(ids), M + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for call in range(2):
        eng.decompose_dev(x.data_ptr(), np.float32, n, len(ids), n, M, rows.data_ptr(), None, None)
        s = eng.summary(len(ids))
        assert eng.fuse_repeats == 0, "the whole call was repeated"
        assert eng.fuse_signal_repairs == len(odd) * (call + 1), "exactly the quantised members are re-run (40, 51, 55 deliver)"
        s2 = eng.summary(len(ids))
        assert all(np.array_equal(s[k], s2[k]) for k in s)
        for j, b in enumerate(ids):
            if call == 1 and b not in (40, 51, 55) and j not in odd:
                continue
            ref = oracle.itd_lean(xs[j], M)
            nr = int(s["n_rows"][j])
            assert nr == ref["rows"].shape[0]
            assert_bits_equal(rows[j, :nr].cpu().numpy(), ref["rows"], "call %d signal %d" % (call, b))
            kc = [int(v) for v in s["knot_counts"][j] if v >= 0]
            if j not in odd:      # (itd_lean counts every level's input plainly; plateau-ridden input follows the reference's NaN-rule counts)
                assert kc[: len(ref["knot_counts"])] == ref["knot_counts"].tolist(), "signal %d knots per level" % b
    eng.close()


def test_fused_call_is_graph_capturable(P, torch, oracle):
    """A decomposition with fused sparse levels has no host synchronisation inside (the verdict is drawn when the summary is read):
    captured into a hipGraph and replayed on new data — a batch over two streams, chunks sharing a knot side — it gives the
    reference's rows; a replay whose data the fused form refuses is repeated level by level by the summary."""
    from pyitd_amd.engine import FUSE_AUTO
    B, n, m = 6, 1 << 17, 6
    x_np = np.stack([sines_noise(n, seed=40 + b, fscale=1 + b / 30.0) for b in range(B)])
    x = torch.from_numpy(x_np).cuda()
    rows = torch.zeros((B, m + 2, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, B, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    eng.set_batch_chunk(1)          # six chunks, two streams, groups of four chunks per knot side
    torch.cuda.synchronize()
    # (the fused levels' workspace is allocated by the first call that takes that path, and a capture cannot allocate: a captured
    #  call on a fresh engine runs level by level — correct, checked below on a second engine — so warm the engine up first)
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
    eng.summary(B)
    cold = P.Engine(n, B, 0)
    cold.set_fuse_min_samples(65536)
    gc = torch.cuda.CUDAGraph()
    side0 = torch.cuda.Stream()
    with torch.cuda.stream(side0):
        with torch.cuda.graph(gc, stream=side0):
            cold.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rows.fill_(float("nan"))
    gc.replay()
    torch.cuda.synchronize()
    s = cold.summary(B)
    ref = oracle.itd_lean(x_np[3], m)
    assert_bits_equal(rows[3, : int(s["n_rows"][3])].cpu().numpy(), ref["rows"], "captured on a fresh engine")
    cold.close()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for seed, with_chirp in ((50, False), (60, True), (70, False)):
        y_np = np.stack([sines_noise(n, seed=seed + b, fscale=1 + b / 30.0) for b in range(B)])
        if with_chirp:
            y_np[2] = coarse(y_np[2])
        x.copy_(torch.from_numpy(y_np))
        rows.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        s = eng.summary(B)
        for b in range(B):
            ref = oracle.itd_lean(y_np[b], m)
            nr = int(s["n_rows"][b])
            assert nr == ref["rows"].shape[0], "replay seed %d signal %d" % (seed, b)
            assert_bits_equal(rows[b, :nr].cpu().numpy(), ref["rows"], "replay seed %d signal %d" % (seed, b))
    assert eng.fuse_repeats >= 1          # the replay with a coarsely quantised member: one of six signals refused (more than one in eight)
    eng.close()


def test_fused_batch_with_odd_members(P, torch, oracle):
    """A float64 batch of ragged length with the caller's baselines buffer under the fused levels: a signal that stops at once, one
    the fused form refuses (re-run on its own, baselines included), then the same batch with a NaN in one input (the reference's NaN
    branch: the whole call is repeated record-driven)."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m, B = (1 << 17) + 333, 6, 10
    xs = np.stack([sines_noise(n, seed=80 + b, fscale=1 + b / 40.0, dtype=np.float64) for b in range(B)])
    xs[6] = np.linspace(-1.0, 1.0, n)
    xs[7] = coarse(xs[7])
    eng = P.Engine(n, B, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    for with_nan in (False, True):
        x = xs.copy()
        if with_nan:
            x[8, n // 3] = np.nan
        xd = torch.from_numpy(x).cuda()
        rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        bs = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        fix0 = eng.fuse_signal_repairs
        eng.decompose_dev(xd.data_ptr(), np.float64, n, B, n, m, rows.data_ptr(), bs.data_ptr(), None)
        s = eng.summary(B)
        if not with_nan:
            assert eng.fuse_signal_repairs - fix0 >= 1 and eng.fuse_repeats == 0
        for b in range(B):
            ref = oracle.itd(x[b], m)
            nr, nb = int(s["n_rows"][b]), int(s["n_baselines"][b])
            assert nr == ref["rows"].shape[0] and nb == ref["baselines"].shape[0], "signal %d" % b
            assert ("natural", "timeout")[int(s["stop"][b])] == ref["stop"]
            assert_bits_equal(rows[b, :nr].cpu().numpy(), ref["rows"], "nan %s signal %d rows" % (with_nan, b))
            assert_bits_equal(bs[b, :nb].cpu().numpy(), ref["baselines"], "nan %s signal %d baselines" % (with_nan, b))
    eng.close()


def test_host_api_through_the_fused_levels(P, oracle, monkeypatch):
    """numpy in, numpy out (ITD().itd, get_baselines) with the fused levels taking every signal of >= 65536 samples: the rows are
    copied to the host only after the summary has drawn the verdict — a refused result (chirp) is repeated level by level first."""
    monkeypatch.setenv("PYITD_FUSE_MIN", "65536")
    d = P.ITD()
    for name, x, m in (("sines", sines_noise(1 << 18, seed=31), 7), ("chirp (refused)", chirp(1 << 17), 5),
                       ("sines again", sines_noise((1 << 17) + 5, seed=32, dtype=np.float64), 9)):
        ref = oracle.itd(x, m)
        assert_bits_equal(d.itd(x, m), ref["rows"], name + ": ITD().itd")
        assert d.stop_reason == ref["stop"]
        assert_bits_equal(d.get_baselines(), ref["baselines"], name + ": get_baselines")


def test_rows_are_final_on_the_stream_without_a_summary(P, torch, oracle):
    """itd_set_valid_flags / itd_set_device_repair (include/pyitd_hip.h): a consumer enqueued on the same stream behind the
    decomposition — here a device-to-device copy of rows_dev — sees the reference's rows with NO itd_get_summary in between, also
    for signals the optimistic forms refuse (a chirp: ties at its extrema; quantised input; a plateau-led signal).  Without the
    repair the validity words say which rows are not final yet."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m = 1 << 17, 6
    lead = np.concatenate([np.zeros(3000), sines_noise(n, seed=11)[3000:].astype(np.float64)]).astype(np.float32)
    xs = np.stack([sines_noise(n, seed=90), chirp(n).astype(np.float32), sines_noise(n, seed=91),
                   (np.round(fuzz_signal(np.random.default_rng(12), 0, n) * 3) / 4.0).astype(np.float32), lead])
    B = len(xs)
    refs = [oracle.itd(x, m) for x in xs]           # (the full restatement: its knot counts follow the reference's NaN rules)
    x = torch.from_numpy(xs).cuda()
    for repair in (False, True):
        eng = P.Engine(n, B, 0)
        eng.set_fuse_mode(FUSE_AUTO)
        eng.set_fuse_min_samples(65536)
        valid = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        eng.set_valid_flags(valid.data_ptr())
        eng.set_device_repair(repair)
        rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        seen = torch.empty_like(rows)
        s = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, s.cuda_stream)
            seen.copy_(rows, non_blocking=True)          # the stream-ordered consumer
            v = valid.clone()
        s.synchronize()
        v = v.cpu().numpy()
        if repair:
            assert v.tolist() == [1] * B
        else:
            assert v[0] == 1 and v[2] == 1 and v[3] == 0 and v[4] == 0, v      # the sines deliver; coarse quantisation and the NaN baseline do not
        for b in range(B):
            if v[b]:
                nr = refs[b]["rows"].shape[0]
                assert_bits_equal(seen[b, :nr].cpu().numpy(), refs[b]["rows"], "repair %s signal %d as the consumer saw it" % (repair, b))
        summ = eng.summary(B)                            # still valid afterwards, and repeats nothing twice
        for b in range(B):
            nr = int(summ["n_rows"][b])
            assert nr == refs[b]["rows"].shape[0]
            assert_bits_equal(rows[b, :nr].cpu().numpy(), refs[b]["rows"], "repair %s signal %d after the summary" % (repair, b))
            kc = [int(k) for k in summ["knot_counts"][b] if k >= 0]
            assert kc[1: 1 + len(refs[b]["knot_counts"])] == refs[b]["knot_counts"].tolist()
        if repair:
            assert eng.device_repairs >= 1 and eng.fuse_repeats == 0 and eng.fuse_signal_repairs == 0
        eng.close()


def test_device_repair_in_a_replayed_graph(P, torch, oracle):
    """The call with its guarded repair captured into a hipGraph: replays on data the fused form delivers and on data it refuses
    leave final rows and validity words with no host involvement at all."""
    from pyitd_amd.engine import FUSE_AUTO
    B, n, m = 3, 1 << 17, 5
    eng = P.Engine(n, B, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    valid = torch.zeros((B,), dtype=torch.int32, device="cuda")
    eng.set_valid_flags(valid.data_ptr())
    eng.set_device_repair(True)
    x = torch.zeros((B, n), dtype=torch.float32, device="cuda")
    rows = torch.zeros((B, m + 2, n), dtype=torch.float64, device="cuda")
    x.copy_(torch.from_numpy(np.stack([sines_noise(n, seed=100 + b) for b in range(B)])))
    torch.cuda.synchronize()
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)      # (allocates the fused levels' workspace)
    eng.summary(B)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for seed, odd in ((110, None), (120, "chirp"), (130, None)):
        y = np.stack([sines_noise(n, seed=seed + b) for b in range(B)])
        if odd:
            y[1] = chirp(n)
        x.copy_(torch.from_numpy(y))
        rows.fill_(float("nan"))
        valid.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert valid.cpu().numpy().tolist() == [1] * B
        for b in range(B):
            ref = oracle.itd_lean(y[b], m)
            assert_bits_equal(rows[b, : ref["rows"].shape[0]].cpu().numpy(), ref["rows"], "replay %d signal %d" % (seed, b))
    eng.close()


@pytest.mark.parametrize("tiles", [16, 32, 64])
def test_small_ranges_and_deep_levels_deliver_or_refuse_never_wrong(P, torch, oracle, tiles):
    """itd_set_fuse_range: many small knot-side workgroups and deep decompositions — knots sparser than a workgroup's range, so the
    halo searches pass many empty workgroups, reach the signal's ends and read the end workgroups' records for the end samples
    alone.  (Round 4's first build re-read an end workgroup it had already passed and doubled a knot: rows wrong from that
    workgroup on with every verification green — the sample pass verifies knots, not table values.  Fixed; the assembled halo is
    now checked for order, and this test holds the families that showed it.)"""
    from pyitd_amd import ITDError
    from pyitd_amd.engine import FUSE_ONLY
    rng = np.random.default_rng(4040 + tiles)
    delivered = 0
    for case in range(18):
        kind = (1, 4, 0, 7, 8, 6)[case % 6]
        n = int(rng.integers(70000, 260000))
        x = sines_noise(n, seed=int(rng.integers(0, 1 << 30))) if kind == 8 else fuzz_signal(rng, kind, n)
        if not np.all(np.isfinite(x)):
            continue
        if kind != 7 and case % 2:
            x = x.astype(np.float32)
        m = (7, 11, 15)[case % 3]
        ref = oracle.itd_lean(x, m)
        eng = P.Engine(n, 1, 0)
        eng.set_fuse_mode(FUSE_ONLY)
        eng.set_fuse_range(tiles)
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        try:
            eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), None, None)
            s = eng.summary(1)
        except ITDError:
            continue                 # refused (capacity, non-finite knot data, verification): reported, not wrong
        finally:
            eng.close()
        nr = int(s["n_rows"][0])
        assert nr == ref["rows"].shape[0], "case %d (family %d, n %d, %d levels)" % (case, kind, n, m + 1)
        assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "case %d (family %d, n %d, %d levels, %d tiles per workgroup)" % (case, kind, n, m + 1, tiles))
        delivered += 1
    assert delivered >= 4


def test_knot_side_with_more_workgroups_than_the_device_holds(P, torch, oracle):
    """A knot-side launch whose workgroups do not all fit the device at once hands out tickets instead of taking its ids from
    blockIdx (itd_knotfirst.hpp): one 2^23-sample signal with 16-tile ranges (1024 workgroups), and a batch of 40 signals of 2^19
    samples in ONE chunk (640 workgroups, several of them for a signal that has stopped) — rows bit-exact, twice in a row (the
    ticket counters clean themselves)."""
    from pyitd_amd.engine import FUSE_ONLY, FUSE_AUTO
    n, m = 1 << 23, 7
    x = sines_noise(n, seed=77)
    ref = oracle.itd_lean(x, m)
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_ONLY)
    eng.set_fuse_range(16)
    xd = torch.from_numpy(x).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for call in range(2):
        rows.fill_(float("nan"))
        torch.cuda.synchronize()
        eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
        s = eng.summary(1)
        nr = int(s["n_rows"][0])
        assert nr == ref["rows"].shape[0]
        assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "2^23 samples, 1024 knot-side workgroups, call %d" % call)
    eng.close()
    del rows, xd
    B, n = 40, 1 << 19
    xs = np.stack([sines_noise(n, seed=300 + b, fscale=1 + b / 64.0) for b in range(B)])
    xs[7] = np.linspace(-1.0, 1.0, n)            # stops at once: its workgroups return at the hand-over
    eng = P.Engine(n, B, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_batch_chunk(B)
    eng.set_batch_streams(1)
    xd = torch.from_numpy(xs).cuda()
    rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for call in range(2):
        eng.decompose_dev(xd.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
        s = eng.summary(B)
        assert eng.fuse_repeats == 0
        for b in (0, 7, 8, 23, 39):
            rb = oracle.itd_lean(xs[b], m)
            nr = int(s["n_rows"][b])
            assert nr == rb["rows"].shape[0], "signal %d" % b
            assert_bits_equal(rows[b, :nr].cpu().numpy(), rb["rows"], "batch of 40 in one chunk, call %d signal %d" % (call, b))
    eng.close()


def test_a_list_that_outgrows_its_workgroup_halves_the_ranges_of_the_next_calls(P, torch, oracle):
    """White noise has ~13 knots per tile at level 3 (830 per 64-tile range: delivered since the hand-over level's layout holds 1720
    candidates) and ~45 at level 2: fused from level 2, with 64 tiles per knot-side workgroup, the candidate lists outgrow the LDS.
    The automatic mode repeats that call level by level and runs the NEXT ones fused with half the tiles per workgroup (a larger
    workspace between two calls; 32-tile ranges hold 1680 knots + the sticky candidates here: at the limit, 16 are safe): every call
    equals the oracle, at most the first two are repeated."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m = 300000, 7
    x = fuzz_signal(np.random.default_rng(77), 0, n).astype(np.float32)
    ref = oracle.itd_lean(x, m)
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    eng.set_fuse_range(0)          # the automatic ranges (a suite run under PYITD_FUSE_RANGE pins them for every other engine)
    eng.set_fuse_level(2)
    xd = torch.from_numpy(x).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    reps = []
    for call in range(4):
        rows.fill_(float("nan"))
        torch.cuda.synchronize()
        eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
        s = eng.summary(1)
        nr = int(s["n_rows"][0])
        assert nr == ref["rows"].shape[0]
        assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "white noise, call %d" % call)
        reps.append(eng.fuse_repeats)
    assert reps[0] == 1 and reps[1] <= 2 and reps[2] == reps[3] == reps[1], "the first call is repeated, then the ranges are halved: %r" % reps
    eng.close()


def test_every_injected_fault_is_refused(P, torch, oracle):
    """The sample pass takes nothing from the knot side on trust (itd_knotfirst.hpp, V0 .. V3): every table entry it uses is checked
    against the tile's own samples (position, value), recomputed (B, S) or tied by bitwise equality to an entry that is; the runs'
    start indices chain; every level's flag words are the exact predicate of the values in the registers.  So ONE perturbed field —
    a value or slope off by an ulp, a position or a start index off by one, a flipped flag bit —, at any fused level (3 .. 8), in an
    interior tile or at a workgroup's range boundary, with ranges of 16 / 32 / 64 tiles, must end in a refusal: > 1000 of them, none
    delivered.  Faults in what a knot-side workgroup RECEIVES from its neighbours (a halo knot's value or position) change what it
    computes: refused too, or — where rounding swallows the perturbation — still the reference's rows, never anything else.  The
    automatic mode then repeats level by level and delivers the oracle's result."""
    from pyitd_amd import ITDError
    from pyitd_amd.engine import FUSE_AUTO, FUSE_ONLY
    n, m, L0 = 1 << 18, 7, 3
    x = sines_noise(n, seed=77)
    ref = oracle.itd_lean(x, m)
    assert ref["rows"].shape[0] == m + 2 and ref["stop"] == "timeout"     # levels 3 .. 8 all run fused
    xd = torch.from_numpy(x).cuda()
    rows = torch.zeros((m + 2, n), dtype=torch.float64, device="cuda")
    n_tiles = n // 512
    rng = np.random.default_rng(2025)
    injected = refused = halo_refused = halo_exact = 0
    for tpw in (16, 32, 64):
        eng = P.Engine(n, 1, 0)
        eng.set_fuse_level(L0)
        eng.set_fuse_range(tpw)
        eng.set_fuse_min_samples(65536)
        eng.set_fuse_mode(FUSE_ONLY)
        eng.set_fuse_cap(-1)                                   # every level fused, whatever PYITD_FUSE_CAP says

        def run():
            rows.zero_()
            torch.cuda.synchronize()                           # (the engine runs on a stream of its own)
            eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
            return eng.summary(1)

        run()                                                  # no fault: delivered
        assert_bits_equal(rows.cpu().numpy(), ref["rows"], "no fault, %d-tile ranges" % tpw)
        for k in range(420):
            kind = int(rng.integers(0, 6))
            level = int(rng.integers(L0, m + 2))               # 3 .. 8
            where = k % 3                                      # a range's first tile, its last tile, an interior tile
            w = int(rng.integers(0, n_tiles // tpw))
            tile = w * tpw + (0 if where == 0 else tpw - 1 if where == 1 else int(rng.integers(1, tpw - 1)))
            slot = int(rng.integers(0, 64))                    # (taken modulo the run's length / the tile's flag words)
            delta = int(rng.choice([1, -1, 2, -3, 1 << 20, -(1 << 30)])) if kind <= 2 else int(rng.choice([1, -1])) if kind <= 4 else int(rng.integers(0, 64))
            eng.debug_kf_fault(kind, level, tile, slot, delta)
            injected += 1
            try:
                run()
            except ITDError as err:
                assert "fused sparse levels" in str(err) and "fail bits" in str(err), str(err)
                refused += 1
                continue
            raise AssertionError("fault not refused: kind %d level %d tile %d (range of %d) slot %d delta %d" % (kind, level, tile, tpw, slot, delta))
        for k in range(12):                                    # the knot side's count of a level's knots (its stop rules' input)
            eng.debug_kf_fault(8, L0 + k % (m + 2 - L0), 0, 0, int(rng.choice([1, -1, 2, -40])))
            injected += 1
            try:
                run()
            except ITDError as err:
                assert "fused sparse levels" in str(err), str(err)
                refused += 1
                continue
            raise AssertionError("a wrong knot count of level %d was not refused" % (L0 + k % (m + 2 - L0)))
        for k in range(60):                                    # what a workgroup receives from its neighbours
            kind, level = 6 + k % 2, int(rng.integers(L0, m + 2))
            w, slot = int(rng.integers(0, n_tiles // tpw)), int(rng.integers(0, 5))
            eng.debug_kf_fault(kind, level, w, slot, int(rng.choice([1, -1, 1 << 25])) if kind == 6 else int(rng.choice([1, -1])))
            try:
                run()
            except ITDError as err:
                assert "fused sparse levels" in str(err), str(err)
                halo_refused += 1
                continue
            assert_bits_equal(rows.cpu().numpy(), ref["rows"], "halo fault delivered: kind %d level %d workgroup %d slot %d" % (kind, level, w, slot))
            halo_exact += 1
        # the automatic mode: the refused call is repeated level by level, the result is the oracle's
        eng.set_fuse_mode(FUSE_AUTO)
        eng.debug_kf_fault(1, 5, n_tiles // 2, 1, 1)
        rep = eng.fuse_repeats
        run()
        assert eng.fuse_repeats == rep + 1
        assert_bits_equal(rows.cpu().numpy(), ref["rows"], "repeated level by level")
        eng.debug_kf_fault(-1)
        eng.set_fuse_mode(FUSE_ONLY)
        run()                                                  # disarmed: delivered again
        assert_bits_equal(rows.cpu().numpy(), ref["rows"], "disarmed")
        eng.close()
    assert injected == refused >= 1000
    assert halo_refused + halo_exact == 180 and halo_refused >= 100


def test_injected_faults_at_the_shipped_hand_over_level(P, torch, oracle):
    """The same proof for the configuration the headline runs: one signal of 2^22 samples, the first fused level chosen by the engine
    (level 2: the hand-over layout — 1720 candidates, the triples in registers —, 64-tile ranges), and 32-tile ranges beside it.  Single
    faults in every field the sample pass reads, at levels 2 .. 8, in range-first, range-last and interior tiles: all refused; the knot
    side's level counts: refused; faults in the halo a workgroup receives — at the hand-over level above all — refused or swallowed by
    rounding, never a different result."""
    from pyitd_amd import ITDError
    from pyitd_amd.engine import FUSE_ONLY
    n, m = 1 << 22, 7
    x = sines_noise(n, seed=123)
    ref = oracle.itd_lean(x, m)
    assert ref["rows"].shape[0] == m + 2 and ref["stop"] == "timeout"
    xd = torch.from_numpy(x).cuda()
    ref_d = torch.from_numpy(ref["rows"]).cuda().view(torch.int64)
    rows = torch.zeros((m + 2, n), dtype=torch.float64, device="cuda")
    n_tiles = n // 512
    rng = np.random.default_rng(606)
    injected = refused = halo_refused = halo_exact = 0
    for tpw, faults in ((0, 1000), (32, 240)):
        eng = P.Engine(n, 1, 0)
        eng.set_fuse_range(tpw)                                # 0 = automatic: 64 tiles
        eng.set_fuse_mode(FUSE_ONLY)
        eng.set_fuse_cap(-1)
        tpw = tpw or 64

        def run():
            rows.zero_()
            torch.cuda.synchronize()
            eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
            return eng.summary(1)

        run()
        assert eng.last_fuse_level == 2, "the automatic first fused level of a 2^22-sample call"
        assert torch.equal(rows.view(torch.int64), ref_d), "no fault"
        L0 = 2
        for k in range(faults):
            kind = int(rng.integers(0, 6))
            level = L0 if k % 4 == 0 else int(rng.integers(L0, m + 2))      # a quarter of them at the hand-over level
            where = k % 3
            w = int(rng.integers(0, n_tiles // tpw))
            tile = w * tpw + (0 if where == 0 else tpw - 1 if where == 1 else int(rng.integers(1, tpw - 1)))
            slot = int(rng.integers(0, 64))
            delta = int(rng.choice([1, -1, 2, -3, 1 << 20, -(1 << 30)])) if kind <= 2 else int(rng.choice([1, -1])) if kind <= 4 else int(rng.integers(0, 64))
            eng.debug_kf_fault(kind, level, tile, slot, delta)
            injected += 1
            try:
                run()
            except ITDError as err:
                assert "fused sparse levels" in str(err) and "fail bits" in str(err), str(err)
                refused += 1
                continue
            raise AssertionError("fault not refused: kind %d level %d tile %d (range of %d) slot %d delta %d" % (kind, level, tile, tpw, slot, delta))
        for k in range(14):
            eng.debug_kf_fault(8, L0 + k % (m + 2 - L0), 0, 0, int(rng.choice([1, -1, 2, -40])))
            injected += 1
            try:
                run()
            except ITDError:
                refused += 1
                continue
            raise AssertionError("a wrong knot count of level %d was not refused" % (L0 + k % (m + 2 - L0)))
        for k in range(120):                                   # the halo as received: two of three at the hand-over level
            kind, level = 6 + k % 2, (L0 if k % 3 else int(rng.integers(L0, m + 2)))
            w, slot = int(rng.integers(0, n_tiles // tpw)), int(rng.integers(0, 5))
            eng.debug_kf_fault(kind, level, w, slot, int(rng.choice([1, -1, 1 << 25])) if kind == 6 else int(rng.choice([1, -1])))
            try:
                run()
            except ITDError as err:
                assert "fused sparse levels" in str(err), str(err)
                halo_refused += 1
                continue
            assert torch.equal(rows.view(torch.int64), ref_d), "halo fault delivered: kind %d level %d workgroup %d slot %d" % (kind, level, w, slot)
            halo_exact += 1
        eng.debug_kf_fault(-1)
        run()
        assert torch.equal(rows.view(torch.int64), ref_d), "disarmed"
        eng.close()
    assert injected == refused >= 1200
    assert halo_refused + halo_exact == 240 and halo_refused >= 120


def test_injected_faults_in_a_later_signal_of_a_batch(P, torch, oracle):
    """... and for batches, which run many signals per launch: eight signals of 2^19 samples (one launch sequence of 2^22: first fused
    level 2), the fault in signal b > 0 (itd_debug_kf_fault_signal).  The automatic mode: the faulted signal — and only it — is refused
    and run again on its own (itd_get_fuse_signal_repairs counts it), the whole batch equals the oracle afterwards; > 1000 faults."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m, B = 1 << 19, 7, 8
    x = np.stack([sines_noise(n, seed=900 + b, fscale=1 + b / 40.0) for b in range(B)])
    refs = [oracle.itd_lean(x[b], m) for b in range(B)]
    assert all(r["rows"].shape[0] == m + 2 for r in refs)
    ref_d = torch.from_numpy(np.stack([r["rows"] for r in refs])).cuda().view(torch.int64)
    xd = torch.from_numpy(x).cuda()
    rows = torch.zeros((B, m + 2, n), dtype=torch.float64, device="cuda")
    n_tiles, tpw, L0 = n // 512, 64, 2
    eng = P.Engine(n, B, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_cap(-1)

    def run():
        rows.zero_()
        torch.cuda.synchronize()
        eng.decompose_dev(xd.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
        s = eng.summary(B)
        assert eng.last_fuse_level == L0
        return s

    run()
    assert (eng.fuse_repeats, eng.fuse_signal_repairs) == (0, 0)
    assert torch.equal(rows.view(torch.int64), ref_d), "no fault"
    rng = np.random.default_rng(707)
    refused = halo_refused = halo_exact = 0
    for k in range(1000):
        kind = int(rng.integers(0, 6)) if k % 25 else 8
        b = 1 + k % (B - 1)
        level = L0 if k % 4 == 0 else int(rng.integers(L0, m + 2))
        where = k % 3
        w = int(rng.integers(0, n_tiles // tpw))
        tile = w * tpw + (0 if where == 0 else tpw - 1 if where == 1 else int(rng.integers(1, tpw - 1)))
        slot = int(rng.integers(0, 64))
        delta = int(rng.choice([1, -1, 2, -3, 1 << 20, -(1 << 30)])) if kind <= 2 else int(rng.choice([1, -1])) if kind in (3, 4, 8) else int(rng.integers(0, 64))
        eng.debug_kf_fault(kind, level, 0 if kind == 8 else tile, slot, delta)
        eng.debug_kf_fault_signal(b)
        before = eng.fuse_signal_repairs
        run()
        assert eng.fuse_repeats == 0
        assert eng.fuse_signal_repairs == before + 1, "fault in signal %d not refused: kind %d level %d tile %d slot %d delta %d" % (b, kind, level, tile, slot, delta)
        assert torch.equal(rows.view(torch.int64), ref_d), "after the repair of signal %d (kind %d level %d)" % (b, kind, level)
        refused += 1
    for k in range(120):
        kind, level = 6 + k % 2, (L0 if k % 3 else int(rng.integers(L0, m + 2)))
        b, w, slot = 1 + k % (B - 1), int(rng.integers(0, n_tiles // tpw)), int(rng.integers(0, 5))
        eng.debug_kf_fault(kind, level, w, slot, int(rng.choice([1, -1, 1 << 25])) if kind == 6 else int(rng.choice([1, -1])))
        eng.debug_kf_fault_signal(b)
        before = eng.fuse_signal_repairs
        run()
        assert eng.fuse_signal_repairs - before in (0, 1)
        halo_refused += eng.fuse_signal_repairs - before
        halo_exact += 1 - (eng.fuse_signal_repairs - before)
        assert torch.equal(rows.view(torch.int64), ref_d), "halo fault in signal %d: kind %d level %d workgroup %d slot %d" % (b, kind, level, w, slot)
    eng.debug_kf_fault(-1)
    eng.debug_kf_fault_signal(0)
    before = eng.fuse_signal_repairs
    run()
    assert eng.fuse_signal_repairs == before and torch.equal(rows.view(torch.int64), ref_d), "disarmed"
    eng.close()
    assert refused == 1000 and halo_refused >= 60 and halo_refused + halo_exact == 120


_R4_CASES = {}


@pytest.mark.parametrize("tiles", [16, 32, 64])
def test_the_inputs_of_round_four_s_wrong_rows(P, torch, oracle, tiles):
    """The eleven draws of the delivery-rate sweep (tools/kf_rates.py, seed 11) on which round 4's first knot-side launch delivered
    wrong rows behind green verifications (helpers.ROUND4_WRONG_ROWS: regenerated from the sweep's seed, not stored): 8 and 12 levels,
    every range size — delivered bit for bit or refused, never anything else."""
    from pyitd_amd import ITDError
    from pyitd_amd.engine import FUSE_ONLY
    if not _R4_CASES:
        for idx, kind, m_, x in kf_rates_draws(11):
            if idx in ROUND4_WRONG_ROWS:
                _R4_CASES[idx] = x
            if idx >= max(ROUND4_WRONG_ROWS):
                break
    assert sorted(_R4_CASES) == sorted(ROUND4_WRONG_ROWS)
    delivered = 0
    for idx, x in sorted(_R4_CASES.items()):
        n = len(x)
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        for m in (7, 11):
            ref = oracle.itd_lean(x, m)
            eng = P.Engine(n, 1, 0)
            eng.set_fuse_mode(FUSE_ONLY)
            eng.set_fuse_range(tiles)
            eng.set_fuse_level(3)                # (the hand-over level of the build that showed it, whatever PYITD_FUSE_LEVEL says)
            eng.set_fuse_cap(-1)
            rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            try:
                eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), None, None)
                s = eng.summary(1)
            except ITDError:
                continue
            finally:
                eng.close()
            nr = int(s["n_rows"][0])
            assert nr == ref["rows"].shape[0], "draw %d, %d levels, %d-tile ranges" % (idx, m + 1, tiles)
            assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "draw %d, %d levels, %d-tile ranges" % (idx, m + 1, tiles))
            delivered += 1
    assert delivered >= 8


def test_batch_pipeline_equals_the_rotating_chunks(P, torch, oracle):
    """itd_set_batch_pipeline(1): the chunks' knot sides in stream order behind their level launches, the sample passes on the engine's
    second stream behind a gate (itd_engine.hip) — the same rows, bit for bit, as the rotating chunks and the oracle; a refusing member
    is run again on its own in either form."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m, B = 1 << 17, 7, 24
    x = np.stack([sines_noise(n, seed=300 + b, fscale=1 + b / 64.0) for b in range(B)])
    x[13] = coarse(x[13])
    refs = [oracle.itd_lean(x[b], m) for b in range(B)]
    xd = torch.from_numpy(x).cuda()
    got = []
    for pipe in (0, 1):
        eng = P.Engine(n, B, 0)
        eng.set_fuse_mode(FUSE_AUTO)
        eng.set_fuse_min_samples(65536)
        eng.set_batch_chunk(5)                       # five chunks, the last one short
        eng.set_batch_pipeline(pipe)
        rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):                           # (twice: the engine's state sets and the gate's counter go on from call to call)
            eng.decompose_dev(xd.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
            s = eng.summary(B)
        assert eng.last_fuse_level >= 2 and eng.fuse_repeats == 0 and eng.fuse_signal_repairs == 2
        for b in range(B):
            nr = refs[b]["rows"].shape[0]
            assert int(s["n_rows"][b]) == nr
            assert_bits_equal(rows[b, :nr].cpu().numpy(), refs[b]["rows"], "pipeline %d, signal %d" % (pipe, b))
        got.append(rows)
        eng.close()


def _run_capped(P, torch, x, m, L0, cap, bases):
    from pyitd_amd.engine import FUSE_ONLY
    n = len(x)
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_ONLY)
    eng.set_fuse_level(L0)
    eng.set_fuse_cap(cap)
    eng.set_fuse_min_samples(65536)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    bs = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda") if bases else None
    torch.cuda.synchronize()
    for _ in range(2):                # (twice: the state sets alternate, the second call starts from what the first one's launches left)
        eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), bs.data_ptr() if bases else None, None)
        took = (eng.last_fuse_level, eng.last_fuse_cap)
        s = eng.summary(1)
    out = {"rows": rows[: int(s["n_rows"][0])].cpu().numpy(), "bases": bs[: int(s["n_baselines"][0])].cpu().numpy() if bases else None,
           "stop": ("natural", "timeout")[int(s["stop"][0])], "knots": [int(v) for v in s["knot_counts"][0] if v >= 0], "took": took}
    eng.close()
    return out


@pytest.mark.parametrize("name,make,m,L0,caps", [
    ("sines 2^19 f32", lambda: sines_noise(1 << 19, seed=41), 7, 2, (4, 6, 8)),
    ("sines 2^18 f64, 12 rows", lambda: sines_noise(1 << 18, seed=42, dtype=np.float64), 10, 3, (5, 9, 11)),
    ("ragged length", lambda: sines_noise((1 << 18) + 1234, seed=43), 7, 3, (5, 7)),
    ("natural stop behind the cap", lambda: sines_noise(1 << 17, seed=6), 20, 3, (5, 8)),
    ("natural stop inside the capped fused levels", lambda: sines_noise(1 << 16, seed=44), 20, 2, (12, 16, 19)),
    ("random walk f32", lambda: fuzz_signal(np.random.default_rng(8), 1, 300000).astype(np.float32), 9, 2, (4, 7)),
])
def test_capped_fused_levels_equal_the_oracle(P, torch, oracle, name, make, m, L0, caps):
    """itd_set_fuse_cap: levels first_fused .. cap - 1 fused, the sample pass leaves the baseline behind level cap - 1, levels cap ..
    max_iteration + 1 one launch each from a scan of it — rows, baselines, knot counts and the stop bit for bit the oracle's, whether
    the signal stops behind the cap, inside the fused levels (the level launches then return at once) or not at all; with the caller's
    baselines buffer and with the engine's rotating slots."""
    from pyitd_amd import ITDError
    x = make()
    ref = oracle.itd(x, m)
    delivered = 0
    for cap in caps:
        for bases in (True, False):
            try:
                got = _run_capped(P, torch, x, m, L0, cap, bases)
            except ITDError:
                continue                       # (deep levels may refuse — reported, as without a cap)
            assert got["took"] == (L0, cap if L0 + 2 <= cap <= m + 1 else 0), (name, cap, got["took"])
            _check(got, ref, "%s, cap %d, %s" % (name, cap, "baselines" if bases else "rows only"))
            delivered += 1
    assert delivered >= 2, name


def test_a_workload_that_fails_at_one_level_keeps_the_levels_in_front_of_it_fused(P, torch, oracle):
    """BASELINE configs[4]'s substitute: the reference's 8000-sample clip tiled — exactly periodic, its baseline collapses to a handful of
    knots at one level, where every sample is a near tie: the fused levels refuse there, every time.  The first call is refused and
    repeated level by level; it leaves the level behind (KfSig::fail_lev) and the calls after it run the levels in front of that one
    fused and the rest level by level: delivered, bit for bit, no further repeat."""
    from pyitd_amd.engine import FUSE_AUTO
    radio = load_golden("radio8000_input")["x"]
    n, m = 1 << 20, 9
    x = np.resize(radio, n).astype(np.float32)
    ref = oracle.itd_lean(x, m)
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    eng.set_fuse_level(0)
    eng.set_fuse_cap(0)
    xd = torch.from_numpy(x).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    forms = []
    for call in range(6):
        rows.fill_(float("nan"))
        torch.cuda.synchronize()
        eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
        forms.append((eng.last_fuse_level, eng.last_fuse_cap))
        s = eng.summary(1)
        nr = int(s["n_rows"][0])
        assert nr == ref["rows"].shape[0] and ("natural", "timeout")[int(s["stop"][0])] == ref["stop"]
        assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "call %d" % call)
        assert [int(v) for v in s["knot_counts"][0] if v >= 0][: len(ref["knot_counts"])] == ref["knot_counts"].tolist(), "call %d" % call
    L0 = forms[0][0]
    assert L0 >= 2 and forms[0][1] == 0, forms                      # the first call: every level fused — refused
    cap = forms[1][1]
    assert cap >= L0 + 2 and all(f == (L0, cap) for f in forms[1:]), forms
    assert eng.fuse_repeats == 1, (eng.fuse_repeats, forms)
    # sixteen delivered capped calls, then one call probes without the cap: refused at the same level, repeated, the cap stays and the
    # next probe is 32 calls away
    for call in range(6, 40):
        eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
        forms.append((eng.last_fuse_level, eng.last_fuse_cap))
        s = eng.summary(1)
    assert [k for k, f in enumerate(forms) if f[1] == 0] == [0, 17], forms
    assert eng.fuse_repeats == 2
    nr = int(s["n_rows"][0])
    assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "call 39")
    # another workload on the same engine: the capped calls deliver it too, and the probe behind them drops the cap
    y = sines_noise(n, seed=77)
    ref_y = oracle.itd_lean(y, m)
    yd = torch.from_numpy(y).cuda()
    torch.cuda.synchronize()
    seen = []
    for call in range(40):
        eng.decompose_dev(yd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
        seen.append(eng.last_fuse_cap)
        s = eng.summary(1)
    assert seen[0] == cap and seen[-1] == 0 and eng.fuse_repeats == 2, seen
    assert_bits_equal(rows[: int(s["n_rows"][0])].cpu().numpy(), ref_y["rows"], "the other workload")
    eng.close()


def test_a_gate_that_gives_up_voids_the_pipelined_call(P, torch, oracle, monkeypatch):
    """The pipelined batch's gate is also what orders a sample pass behind its own knot side; one that gives up (here: a time-out of zero,
    PYITD_PIPE_GATE_US=0) says so, and the call is void: refused as a whole, repeated level by level — the oracle's rows —, and the engine's
    later batches rotate over the streams again.  With the repair on the device (itd_set_device_repair) the same holds without the host."""
    from pyitd_amd.engine import FUSE_AUTO
    monkeypatch.setenv("PYITD_PIPE_GATE_US", "0")
    n, m, B = 1 << 17, 6, 12
    x = np.stack([sines_noise(n, seed=500 + b, fscale=1 + b / 64.0) for b in range(B)])
    refs = [oracle.itd_lean(x[b], m) for b in range(B)]
    xd = torch.from_numpy(x).cuda()
    for device_repair in (False, True):
        eng = P.Engine(n, B, 0)
        eng.set_fuse_mode(FUSE_AUTO)
        eng.set_fuse_min_samples(65536)
        eng.set_batch_chunk(4)
        eng.set_batch_pipeline(1)
        valid = torch.zeros(B, dtype=torch.int32, device="cuda")
        if device_repair:
            eng.set_valid_flags(valid.data_ptr())
            eng.set_device_repair(True)
        rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for call in range(3):
            rows.fill_(float("nan"))
            torch.cuda.synchronize()
            eng.decompose_dev(xd.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
            s = eng.summary(B)
            for b in range(B):
                nr = refs[b]["rows"].shape[0]
                assert int(s["n_rows"][b]) == nr
                assert_bits_equal(rows[b, :nr].cpu().numpy(), refs[b]["rows"], "call %d, signal %d, device repair %s" % (call, b, device_repair))
        if device_repair:
            assert eng.device_repairs >= B and bool((valid == 1).all())
        else:
            assert eng.fuse_repeats == 1          # the first call only: the engine left the pipelined form behind
        eng.close()


def test_a_captured_fused_call_survives_a_workspace_change(P, torch, oracle):
    """A hipGraph that holds a fused call has the fused levels' workspace pointers baked into its launches.  When later calls of the
    same engine need a larger workspace (smaller ranges: itd_set_fuse_range, or the automatic halving after a capacity refusal) the
    old one is retired, not freed: the graph replays correctly afterwards, and so do direct calls."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m = 1 << 18, 7
    x_np = sines_noise(n, seed=91)
    x = torch.from_numpy(x_np).cuda()
    rows = torch.zeros((m + 2, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    eng.set_fuse_range(64)
    torch.cuda.synchronize()
    eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)      # allocates the 64-tile workspace
    eng.summary(1)
    ws0 = eng.workspace_bytes
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    eng.set_fuse_range(16)                                      # four times the workgroups: a new, larger workspace
    rows.fill_(float("nan"))
    eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
    eng.summary(1)
    assert eng.workspace_bytes > ws0 and eng.fuse_repeats == 0
    ref = oracle.itd_lean(x_np, m)
    assert_bits_equal(rows.cpu().numpy(), ref["rows"], "direct call with 16-tile ranges")
    for seed in (92, 93):                                       # the graph still refers to the retired workspace
        y_np = sines_noise(n, seed=seed)
        x.copy_(torch.from_numpy(y_np))
        rows.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        s = eng.summary(1)
        assert eng.fuse_repeats == 0
        assert_bits_equal(rows[: int(s["n_rows"][0])].cpu().numpy(), oracle.itd_lean(y_np, m)["rows"], "replay after the change, seed %d" % seed)
    eng.close()


def test_a_workload_the_fused_form_cannot_deliver_backs_off_exponentially(P, torch, oracle):
    """A coarsely quantised signal refuses the fused levels on every attempt.  The automatic mode repeats the refused call level by
    level and runs the next 16 calls that way, then tries again — and doubles the pause every time the attempt behind a pause refuses
    (32, 64, ... 1024): over 60 calls that is three wasted attempts (calls 1, 18, 51), not four; every call's rows are the oracle's;
    itd_get_last_fuse_level says which form a call took; a workload that delivers starts over at 16."""
    from pyitd_amd.engine import FUSE_AUTO
    n, m = 1 << 17, 6
    x = coarse(sines_noise(n, seed=101))
    ref = oracle.itd_lean(x, m)
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    eng.set_fuse_level(0)             # automatic (a suite run under PYITD_FUSE_LEVEL pins it for every other engine): level 3 at this size
    xd = torch.from_numpy(x).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    attempts = []
    for call in range(1, 61):
        eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
        form = eng.last_fuse_level
        s = eng.summary(1)
        if form:
            attempts.append(call)
            assert form == 3 and eng.last_fuse_level == 0          # (the summary repeated it level by level)
        if call in (1, 2, 18, 19, 51, 60):
            nr = int(s["n_rows"][0])
            assert nr == ref["rows"].shape[0]
            assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "call %d" % call)
    assert attempts == [1, 18, 51] and eng.fuse_repeats == 3
    y = sines_noise(n, seed=102)                                    # a signal the fused form delivers: the pause starts over
    yd = torch.from_numpy(y).cuda()
    torch.cuda.synchronize()
    eng.set_fuse_mode(FUSE_AUTO)                                    # (clears the pause that is still running)
    eng.decompose_dev(yd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
    assert eng.last_fuse_level == 3
    eng.summary(1)
    assert eng.fuse_repeats == 3 and eng.last_fuse_level == 3
    eng.close()
