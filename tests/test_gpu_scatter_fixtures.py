"""The pad-plane scatter and the track integrator of the HIP library against the fixtures the
reference's own code generated (tests/golden/transport.npz, tracks.npz) -- one hop, no oracle in
between -- and the cases only explicit samples can reach (a time bucket with more lit pads than the LDS
table holds).  Needs a real MI355X: ``-m gpu``.

Tolerances: keys, labels, zero-charge inserts exact; charges <= 2 electrons (numpy's exp in the
reference's pdf vs the kernel's constant weight table; same bound as oracle <-> golden)."""
import ctypes as C

import numpy as np
import pytest

from attpc_engine_amd import GasTarget, _abi, nuclear_map, workloads
from attpc_engine_amd.detector.luts import build_det_desc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return _abi.Context(0)


@pytest.fixture(scope="module")
def orc():
    from oracle import pyoracle
    return pyoracle


def _configure(ctx, diffusion, fold=True, **det_kw):
    gas = GasTarget([(1, 2, 2)], 300.0, nuclear_map)
    cfg = workloads.detector_config(gas, diffusion=diffusion)
    for k, v in det_kw.items():
        setattr(cfg.det_params, k, v)
    desc, keep = build_det_desc(cfg, [nuclear_map.get_data(1, 1)], fold_beam=True)
    ctx._det_token = None  # configured through the C ABI directly: the shim's cache no longer describes the device
    ctx.check(ctx.lib.attpc_det_configure(ctx.handle, desc), "attpc_det_configure")
    raw, keep_raw = build_det_desc(cfg, [nuclear_map.get_data(1, 1)], fold_beam=False)
    return cfg, raw, (keep, keep_raw)


def device_scatter(ctx, events, seed=11, first_event=0, n_rows=18):
    """events: list of events, each a list of (xyt [n,3], electrons [n], label) in insertion order.
    -> per event (points [P,3], labels [P]) through attpc_det_scatter."""
    n_sim = max(len(ev) for ev in events)
    lay = _abi.EventLayout()
    lay.n_rows, lay.n_sim = n_rows, n_sim
    for i in range(_abi.MAX_ROWS):
        lay.species_of_row[i] = -1
    labels = [lab for _, _, lab in events[0]] + [0] * (n_sim - len(events[0]))
    for ev in events:  # one layout for the launch: every event uses the same label per position
        assert [lab for _, _, lab in ev] == labels[:len(ev)]
    for i, lab in enumerate(labels):
        lay.indices[i] = int(lab)
    counts, rows = [], []
    for ev in events:
        for i in range(n_sim):
            if i < len(ev):
                xyt, el, _ = ev[i]
                counts.append(len(xyt))
                rows.append(np.column_stack([np.asarray(xyt, dtype=np.float64), np.asarray(el, dtype=np.float64)]))
            else:
                counts.append(0)
    samples = np.ascontiguousarray(np.concatenate(rows)) if rows else np.zeros((0, 4))
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n = len(events)
    capacity = 1 << 16
    while True:
        offsets = np.zeros(n + 1, dtype=np.int64)
        points = np.empty((capacity, 3))
        lab = np.empty(capacity, dtype=np.int64)
        out = _abi.CloudOut(capacity, _abi.iptr(offsets, C.c_int64), _abi.dptr(points), _abi.iptr(lab, C.c_int64), None)
        stats = _abi.RunStats()
        rc = ctx.lib.attpc_det_scatter(ctx.handle, seed, first_event, n, lay, _abi.dptr(samples),
                                       _abi.iptr(counts, C.c_int32), out, stats)
        if rc == _abi.E_CAPACITY:
            capacity = int(stats.n_points) + 16
            continue
        ctx.check(rc, "attpc_det_scatter")
        break
    return [(points[offsets[e]:offsets[e + 1]].copy(), lab[offsets[e]:offsets[e + 1]].copy()) for e in range(n)], stats.as_dict()


def _sorted(pad, tb, charge, label):
    order = np.lexsort((tb, pad))
    return pad[order], tb[order], charge[order], label[order]


def _compare_with_dict(points, labels, tb_ref, pad_ref, charge_ref, label_ref):
    """Device cloud vs a reference dictionary (keys in any order): the 0 <= tb < 512 mask of
    simulator.py:111-113 applied to the reference side."""
    keep = (tb_ref >= 0) & (tb_ref < 512) & (pad_ref >= 0)
    ref = _sorted(pad_ref[keep], tb_ref[keep], charge_ref[keep], label_ref[keep])
    tb_dev = np.floor(points[:, 1]).astype(np.int64)
    assert ((points[:, 1] - tb_dev) >= 0).all() and ((points[:, 1] - tb_dev) < 1).all()  # jitter in [0, 1)
    dev = _sorted(points[:, 0].astype(np.int64), tb_dev, points[:, 2].astype(np.int64), labels)
    np.testing.assert_array_equal(dev[0], ref[0])
    np.testing.assert_array_equal(dev[1], ref[1])
    np.testing.assert_array_equal(dev[3], ref[3])
    diff = np.abs(dev[2] - ref[2])
    assert diff.max(initial=0) <= 2, diff.max()
    assert ((dev[2] == 0) == (ref[2] == 0)).all()  # zero-charge inserts kept, nothing else is zero
    return int(diff.max(initial=0)), float((diff == 0).mean()) if len(diff) else 1.0


@pytest.mark.parametrize("name", ["mixed", "overlap", "nodiffusion", "bigdiffusion"])
def test_scatter_kernel_vs_reference_transport(golden_dir, ctx, name):
    """transport_track / pairing / dict semantics of the REFERENCE (transporter.py:172-317 executed on
    its own 5600x5600 grid by tests/golden/make_golden.py) straight against scatter_kernel: beam pads,
    off-grid pixels, zero diffusion (point_transport), 10x diffusion, overlapping labels (last writer
    wins), zero-charge inserts."""
    g = np.load(golden_dir / "transport.npz")
    _configure(ctx, float(g[f"{name}_diffusion"]))
    cases = [(g[f"{name}_xyt{i}"], g[f"{name}_electrons{i}"], int(g[f"{name}_label{i}"]))
             for i in range(int(g[f"{name}_n_cases"]))]
    for variant in (1, 2, 3):  # the three builds of the kernel (u32 sums: two / one workgroup per CU; u64 sums)
        ctx.set_option("scatter_variant", variant)
        (cloud,), stats = device_scatter(ctx, [cases])
        tbpad = g[f"{name}_tbpad"]
        worst, exact = _compare_with_dict(cloud[0], cloud[1], tbpad[:, 0], tbpad[:, 1], g[f"{name}_charge"], g[f"{name}_labels"])
        assert exact > 0.999 and stats["n_failed"] == 0 and stats["n_inconsistent"] == 0
        print(name, "variant", variant, "points", len(cloud[0]), "max |dq|", worst, "exact fraction", exact)
    ctx.set_option("scatter_variant", 0)
    if name == "mixed":  # dict_to_points (simulator.py:19-49) of the reference on the same dictionary
        pa = g["mixed_point_array"]
        _compare_with_dict(cloud[0], cloud[1], pa[:, 1].astype(np.int64), pa[:, 0].astype(np.int64),
                           pa[:, 2].astype(np.int64), g["mixed_label_array"])


def test_track_kernel_vs_reference_radau(golden_dir, ctx):
    """generate_trajectory + generate_electrons (Fano 0) of the reference (solver.py:243-347, scipy Radau
    at rtol 1e-3, tests/golden/tracks.npz: 14 tracks p/d/alpha/10Be/12C/16O, 0.3-60 MeV) straight
    against track_kernel: recorded rows within max(3, 2 %), positions within 0.5 mm, running electron
    sum within 3 % + 5 (the per-sample counts of the reference carry its dense-output noise)."""
    from tests.test_oracle_golden import _golden_det

    g = np.load(golden_dir / "tracks.npz")
    det, keep = _golden_det(g, fano=0.0)
    ctx._det_token = None  # configured through the C ABI directly: the shim's cache no longer describes the device
    ctx.check(ctx.lib.attpc_det_configure(ctx.handle, det), "attpc_det_configure")
    species = [tuple(s) for s in g["species"]]
    n = len(g["cases"])
    lay = _abi.EventLayout()
    lay.n_rows, lay.n_sim = 4, 1
    lay.indices[0] = 2
    checked_positions = 0
    for i, case in enumerate(g["cases"]):
        for r in range(_abi.MAX_ROWS):
            lay.species_of_row[r] = -1
        lay.species_of_row[2] = species.index((int(case[0]), int(case[1])))
        p4 = np.zeros((1, 4, 4))
        p4[0, 2] = g[f"mom{i}"]
        vertex = np.ascontiguousarray(case[5:8][None, :])
        samples = np.zeros((1, _abi.TIME_SAMPLES, 4))
        counts = np.empty(1, dtype=np.int32)
        steps = np.empty(1, dtype=np.int32)
        ctx.check(ctx.lib.attpc_det_tracks(ctx.handle, 1, 0, 1, lay, _abi.dptr(p4), _abi.dptr(vertex), _abi.TIME_SAMPLES,
                                           _abi.dptr(samples), _abi.iptr(counts, C.c_int32), _abi.iptr(steps, C.c_int32)),
                  "attpc_det_tracks")
        n_ref = int(g[f"nrows{i}"])
        assert abs(int(steps[0]) - n_ref) <= max(3, 0.02 * n_ref), (i, steps[0], n_ref)
        kept = samples[0, : counts[0]]
        gain = det.mpgd_gain
        el = kept[:, 3] / gain
        head = g[f"electrons_head{i}"]
        # Row 0 never makes electrons (solver.py:338-339) and the only other samples without any are the
        # last ones of a particle that ranges out (oracle: a contiguous tail), so kept sample s is ODE row s + 1
        ref = g[f"track{i}"]  # every 10th row of the reference's track
        rows = np.arange(10, min(int(counts[0]) + 1, 10 * len(ref)), 10)
        if len(rows):
            mine = kept[rows - 1]
            dv = det.length / (det.windows_edge - det.micromegas_edge)
            z = det.length - (mine[:, 2] - det.micromegas_edge) * dv
            ref_rows = ref[rows // 10]
            assert np.abs(mine[:, 0] - ref_rows[:, 0]).max() < 5e-4 and np.abs(mine[:, 1] - ref_rows[:, 1]).max() < 5e-4
            assert np.abs(z - ref_rows[:, 2]).max() < 5e-4
            m = min(len(head) - 1, len(el), 40)
            cum_err = np.abs(np.cumsum(el[:m]) - np.cumsum(head[1:m + 1])).max()
            assert cum_err <= 0.03 * head[1:m + 1].sum() + 5, (i, cum_err)
            checked_positions += 1
        ref_sum = int(g[f"electrons_sum{i}"])
        assert abs(int(el.sum()) - ref_sum) <= max(30, 0.01 * ref_sum), (i, int(el.sum()), ref_sum)
    assert checked_positions >= 13, checked_positions


def test_reference_tracks_through_scatter_vs_oracle(golden_dir, ctx, orc):
    """The reference's own Radau tracks + Fano-0 electrons (tracks.npz), turned into drift samples as
    generate_point_cloud does (solver.py:387-398), through scatter_kernel: equal to the oracle's
    transport of the same samples."""
    g = np.load(golden_dir / "tracks.npz")
    cfg, raw, keep = _configure(ctx, 0.277)
    dv = cfg.drift_velocity
    events = []
    for i in range(len(g["cases"])):
        track = g[f"track{i}"]
        head = g[f"electrons_head{i}"]
        m = min(len(track), 7)  # rows 0, 10, 20 ... of the track with the electron counts of rows 0..6 (Fano 0)
        el = head[:m].astype(np.int64)
        keep_rows = el >= 1
        xyt = np.column_stack([track[:m, 0], track[:m, 1], (cfg.det_params.length - track[:m, 2]) / dv + 10.0])[keep_rows]
        events.append([(xyt, el[keep_rows] * cfg.det_params.mpgd_gain, 2)])
    clouds, stats = device_scatter(ctx, events)
    total = 0
    for ev, (pts, lab) in zip(events, clouds):
        keys, charge, labels = orc.transport(raw, ev)
        tb = np.empty(len(keys), dtype=np.int64)
        pad = np.empty(len(keys), dtype=np.int64)
        for k, key in enumerate(keys):
            t, p = orc.unpair(int(key))
            tb[k], pad[k] = t, p
        _compare_with_dict(pts, lab, tb, pad, charge, labels)
        total += len(pts)
    assert total > 500 and stats["n_failed"] == 0


def _plane_filling_event(cfg, n_tracks=4, pitch_mm=4.0, tb=500.25, electrons=3_000_000):
    """Samples on a regular grid over the whole pad plane, ALL in one time bucket."""
    xs = np.arange(-270.0, 270.0, pitch_mm)
    gx, gy = np.meshgrid(xs, xs, indexing="ij")
    inside = gx ** 2 + gy ** 2 < 268.0 ** 2
    pts = np.column_stack([gx[inside] * 1e-3, gy[inside] * 1e-3])
    rng = np.random.default_rng(4)
    rng.shuffle(pts)
    parts = np.array_split(pts, n_tracks)
    labels = [2, 4, 6, 7][:n_tracks]
    return [(np.column_stack([p, np.full(len(p), tb)]), np.full(len(p), electrons, dtype=np.int64), lab)
            for p, lab in zip(parts, labels)]


@pytest.mark.parametrize("variant,must_be_lone", [(1, True), (3, True), (2, False)])
def test_lone_time_bucket_larger_than_the_lds_table(ctx, orc, variant, must_be_lone):
    """One time bucket of one event lights > 8192 pads -- more than the 6144-slot table of the two-workgroup build (1)
    and the 8192-slot table of the u64 build (3) hold.  The reference's dict has no limit (simulator.py:93-101); the
    bucket goes through lone_bucket_kernel's direct-mapped table and the cloud equals the oracle's, beside ordinary
    events in the same launch.  (The one-workgroup build (2) has 12 288 slots since round 3, more than the pad plane has
    pads: it may hold the bucket; the cloud is checked either way.)"""
    cfg, raw, keep = _configure(ctx, 0.277)
    big = _plane_filling_event(cfg)
    small = [(xyt[:40] * np.array([1.0, 1.0, 0.5]), el[:40], lab) for xyt, el, lab in big]  # tb 250: an ordinary event
    ctx.set_option("scatter_variant", variant)
    try:
        clouds, stats = device_scatter(ctx, [small, big, small])
    finally:
        ctx.set_option("scatter_variant", 0)
    assert stats["n_failed"] == 0 and stats["n_inconsistent"] == 0
    assert stats["n_lone_buckets"] >= 1 or not must_be_lone
    for ev, (pts, lab) in zip([small, big, small], clouds):
        keys, charge, labels = orc.transport(raw, ev)
        tb, pad = np.array([orc.unpair(int(k)) for k in keys], dtype=np.int64).T
        _compare_with_dict(pts, lab, tb, pad, charge, labels)
    lit = len(clouds[1][0])
    assert lit > 8192, lit
    print("variant", variant, "pads lit in the lone bucket:", lit, "lone buckets:", stats["n_lone_buckets"])


def test_lone_time_bucket_with_monte_carlo_diffusion(ctx, orc):
    """The same overflow with the per-electron Monte-Carlo diffusion extension (lone_bucket_kernel<true>): one
    track (the oracle's diagnostic entry numbers the entries of a single track like the kernel does),
    seed 0 / event 0 as orc_transport_track uses them."""
    cfg, raw, keep = _configure(ctx, 0.277, mc_diffusion=True)
    (xyt, el, lab), = _plane_filling_event(cfg, n_tracks=1, pitch_mm=4.8, electrons=20 * 175000)
    assert len(xyt) <= 10112
    ev = [(xyt, el, lab)]
    ctx.set_option("scatter_variant", 1)  # the 6144-slot table (the automatic choice for this extension has 12 288 slots)
    try:
        clouds, stats = device_scatter(ctx, [ev], seed=0)
    finally:
        ctx.set_option("scatter_variant", 0)
    assert stats["n_failed"] == 0 and stats["n_inconsistent"] == 0
    assert stats["n_lone_buckets"] >= 1 or len(clouds[0][0]) <= 6144, (stats["n_lone_buckets"], len(clouds[0][0]))
    keys, charge, labels = orc.transport(raw, ev)
    tb, pad = np.array([orc.unpair(int(k)) for k in keys], dtype=np.int64).T
    pts, lab_dev = clouds[0]
    keep_ref = (tb >= 0) & (tb < 512)
    order_ref = np.lexsort((tb[keep_ref], pad[keep_ref]))
    order_dev = np.lexsort((np.floor(pts[:, 1]), pts[:, 0]))
    np.testing.assert_array_equal(pts[order_dev, 0].astype(np.int64), pad[keep_ref][order_ref])
    np.testing.assert_array_equal(pts[order_dev, 2].astype(np.int64), charge[keep_ref][order_ref])  # whole electrons: exact
    assert len(pts) > 4096
    print("pads lit:", len(pts), "lone buckets:", stats["n_lone_buckets"])
    _configure(ctx, 0.277)


def test_pad_ids_outside_the_key_range_are_rejected(ctx):
    """a custom pad grid with ids >= 16384 (or < -1) must not corrupt keys silently."""
    gas = GasTarget([(1, 2, 2)], 300.0, nuclear_map)
    cfg = workloads.detector_config(gas)
    grid = cfg.pad_grid.copy().astype(np.int64)
    grid[100, 100] = 20000
    cfg.pad_grid = grid
    with pytest.raises(ValueError, match="pad ids"):
        build_det_desc(cfg, [nuclear_map.get_data(1, 1)])
    cfg.pad_grid = np.where(grid == 20000, -2, grid)
    with pytest.raises(ValueError, match="pad ids"):
        build_det_desc(cfg, [nuclear_map.get_data(1, 1)])
    # and the C ABI itself refuses such a table
    good_cfg = workloads.detector_config(gas)
    desc, keep = build_det_desc(good_cfg, [nuclear_map.get_data(1, 1)])
    lut = np.ctypeslib.as_array(desc.pad_lut, shape=(desc.lut_n * desc.lut_n,)).copy()
    lut[5] = 16384
    desc.pad_lut = lut.ctypes.data_as(C.POINTER(C.c_int16))
    with pytest.raises(ValueError, match="pad id"):
        ctx._det_token = None  # configured through the C ABI directly: the shim's cache no longer describes the device
        ctx.check(ctx.lib.attpc_det_configure(ctx.handle, desc), "attpc_det_configure")


def test_rows_too_large_for_the_transfer_record_go_the_plain_way(ctx, orc):
    """Charges of 2^45 electrons and more do not fit the 16-byte transfer record (include/attpc_engine.h): such a
    chunk is delivered in the reference's dtypes instead -- same rows either way.  (Also the u64 slow path of the
    scatter for pixels above 2^28 electrons.)"""
    cfg, raw, keep = _configure(ctx, 0.277)
    xyt = np.array([[0.05, 0.04, 300.5], [0.051, 0.041, 300.7], [-0.1, 0.02, 120.2]])
    el = np.array([5_000_000_000_000_000, 3_000_000, 9_000_000_000_000_000], dtype=np.int64)  # exact in f64
    ev = [(xyt, el, 2)]
    keys, charge, labels = orc.transport(raw, ev)
    assert charge.max() >= (1 << 45)
    tb, pad = np.array([orc.unpair(int(k)) for k in keys], dtype=np.int64).T
    for compact in (1, 0):
        ctx.set_option("compact_transfer", compact)
        try:
            (cloud,), stats = device_scatter(ctx, [ev])
        finally:
            ctx.set_option("compact_transfer", 1)
        # the per-pixel truncation of products this large may differ in the last bits of a 2^52-sized number
        keep_ref = (tb >= 0) & (tb < 512)
        o_ref = np.lexsort((tb[keep_ref], pad[keep_ref]))
        o_dev = np.lexsort((np.floor(cloud[0][:, 1]), cloud[0][:, 0]))
        np.testing.assert_array_equal(cloud[0][o_dev, 0].astype(np.int64), pad[keep_ref][o_ref])
        np.testing.assert_allclose(cloud[0][o_dev, 2], charge[keep_ref][o_ref].astype(np.float64), rtol=1e-12)
        assert stats["n_failed"] == 0
