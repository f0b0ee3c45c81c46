"""Host logic of the geometric march, checked WITHOUT a GPU (lf_march_tables): the device does not
walk the selected ghost paths one by one but as ONE tree per wavelength -- shared legs once, forks
parked and restored, dead waves jumping over what only they would visit.  This test interprets the
program the way a wave that never loses a ray would, and checks that
  * every path comes out, exactly once, with exactly the rows of its own flat sequence
    (backwards N-1..i+1, mirror at i, forwards i+1..j-1, mirror at j, backwards j-1..0);
  * a row's multiplicity is the number of paths that run through it (events and ray fates are
    tallied per path from it);
  * the jump taken by a dead wave lands where, and with the parked state with which, a live wave
    would arrive after finishing everything that only the dead rays would still have visited;
  * runs of plain rows have one multiplicity, never continue past a path's last row, and only start
    with a curved mirror or a plain row."""
import ctypes as C

import numpy as np
import pytest

MIRROR, STOP, FLAT, REST1, SAVE0, SAVE1, END, REST0 = 1, 2, 4, 8, 0x10, 0x20, 0x40, 0x80
MAX_PAIRS = 128


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def tables(pkg, lens, pairs=None, primary=True):
    lib = pkg.load_library()
    n, nl = int(lens["n"]), lens["ior"].shape[0]
    f = lambda a: np.ascontiguousarray(a, np.float32)   # noqa: E731
    rad, thk, ior, sa = f(lens["radius"]), f(lens["thickness"]), f(lens["ior"]), f(lens["semi_aperture"])
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))   # noqa: E731
    info = np.zeros(8 + 4 * (MAX_PAIRS + 1), np.int32)
    ip = info.ctypes.data_as(C.POINTER(C.c_int))
    pa = None if pairs is None else np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    pp = None if pa is None else pa.ctypes.data_as(C.POINTER(C.c_int))
    args = [n, int(lens["stop"]), nl, fp(rad), fp(thk), fp(ior), fp(sa), pp, 0 if pa is None else len(pa), int(primary), ip]
    assert lib.lf_march_tables(*args, None, C.c_size_t(0), None, C.c_size_t(0)) == 0
    rows = np.zeros((info[4], 8), np.float32)
    skip = np.zeros(info[5], np.int32)
    assert lib.lf_march_tables(*args, fp(rows), C.c_size_t(rows.size), skip.ctypes.data_as(C.POINTER(C.c_int)),
                               C.c_size_t(skip.size)) == 0
    n_paths, flat_rows, prog_off, prog_rows = (int(v) for v in info[:4])
    paths = [tuple(int(v) for v in info[8 + 4 * q:12 + 4 * q]) for q in range(n_paths)]
    return dict(rows=rows, flags=rows.view(np.int32)[:, 5].astype(np.int64) & 0xffffffff, skip=skip, paths=paths,
                flat_rows=flat_rows, prog_off=prog_off, prog_rows=prog_rows, n_lambda=nl)


def geometry(t, r):
    """what the event arithmetic reads from a row (everything but the bookkeeping flags)"""
    return tuple(t["rows"][r][[0, 1, 2, 3, 4, 6, 7]].tolist()) + (int(t["flags"][r]) & 7,)


def check(t):
    for lam in range(t["n_lambda"]):
        p0 = t["prog_off"] + lam * t["prog_rows"]
        fl = t["flags"][p0:p0 + t["prog_rows"]]
        mult = (fl >> 16) & 0xff
        run = (fl >> 8) & 0xff
        # ---- a live wave's walk ------------------------------------------------------------
        seq, slots, done = [], {0: None, 1: None}, {}
        before = {}          # program row -> (sequence before it, slot snapshots)
        r = 0
        while r < t["prog_rows"]:
            before[r] = (list(seq), {k: (None if v is None else list(v)) for k, v in slots.items()})
            if fl[r] & SAVE0: slots[0] = list(seq)
            if fl[r] & SAVE1: slots[1] = list(seq)
            seq.append(r)
            if fl[r] & END:
                q = int(fl[r] >> 24)
                assert q not in done, "a path completes twice"
                done[q] = list(seq)
                if fl[r] & REST1: seq = list(slots[1])
                elif fl[r] & REST0: seq = list(slots[0])
                else: assert r == t["prog_rows"] - 1, "only the last row may end the walk"
            r += 1
        before[t["prog_rows"]] = (None, None)
        # ---- every path, once, with its own rows -------------------------------------------
        assert sorted(done) == list(range(len(t["paths"])))
        for q, (i, j, off, cnt) in enumerate(t["paths"]):
            flat0 = lam * t["flat_rows"] + off
            want = [geometry(t, flat0 + k) for k in range(cnt)]
            got = [geometry(t, p0 + r) for r in done[q]]
            assert got == want, (lam, q, i, j)
        # ---- multiplicity = paths through the row -------------------------------------------
        through = np.zeros(t["prog_rows"], np.int64)
        for rows_q in done.values():
            through[rows_q] += 1
        assert np.array_equal(through, mult)
        # ---- runs ----------------------------------------------------------------------------
        for r in range(t["prog_rows"]):
            if run[r] == 0:
                continue
            first = fl[r] & 0xff
            assert first & ~(MIRROR | SAVE0 | SAVE1) == 0 or first & ~(END | REST0 | REST1) == 0
            for k in range(1, run[r]):
                assert fl[r + k] & (MIRROR | STOP | FLAT | SAVE0 | SAVE1) == 0    # plain rows follow
                assert mult[r + k] == mult[r]
                assert not fl[r + k - 1] & END                                    # nothing after an END
        # ---- the jump of a dead wave ------------------------------------------------------------
        for r in range(t["prog_rows"]):
            sk = int(t["skip"][r])          # the same table serves every wavelength
            tgt, rs = r + (sk >> 2), sk & 3
            assert r < tgt <= t["prog_rows"]
            mine = set(q for q, rows_q in done.items() if r in rows_q)
            # everything skipped is visited only by paths through r ...
            for s in range(r + 1, tgt):
                assert set(q for q, rows_q in done.items() if s in rows_q) <= mine, (r, s)
            # ... and the landing row is not (or the program is over)
            if tgt < t["prog_rows"]:
                assert not set(q for q, rows_q in done.items() if tgt in rows_q) <= mine, (r, tgt)
                want_seq = before[tgt][0]
                parked = before[r + 1][1] if r + 1 in before and before[r + 1][1] else before[r][1]
                # state the jump restores: the slot as it is right after row r executed
                slots_after = dict(before[r][1])
                if fl[r] & SAVE0: slots_after[0] = before[r][0]
                if fl[r] & SAVE1: slots_after[1] = before[r][0]
                assert rs in (1, 2)
                assert slots_after[0 if rs == 2 else 1] == want_seq, (r, tgt, rs)
                del parked
            else:
                assert rs == 0 or tgt == t["prog_rows"]


def test_double_gauss_all_pairs(pkg):
    lens = pkg.load_lens_file("dgauss11.lens")
    t = tables(pkg, lens)
    assert len(t["paths"]) == 46 and t["flat_rows"] == 886 and t["prog_rows"] == 426
    check(t)


def test_odd_selections(pkg):
    lens = pkg.load_lens_file("dgauss11.lens")
    for pairs, primary in [([(3, 8), (0, 2), (3, 4), (0, 10), (3, 8), (7, 9), (0, 2)], True),
                           ([(2, 9)], False), ([(-1, -1)], False),
                           ([(0, 10), (1, 10), (2, 10), (9, 10)], True), ([(4, 6), (4, 7), (4, 10)], False),
                           ([(i, j) for i in range(5) for j in range(i + 1, 5)], False)]:
        check(tables(pkg, lens, pairs, primary))


def test_other_prescriptions(pkg):
    check(tables(pkg, pkg.load_lens_file("thinlens.lens")))
    flat = dict(n=5, stop=2, radius=np.array([45.0, 0.0, 0.0, 0.0, -38.0], np.float32),
                thickness=np.array([6.0, 4.0, 4.0, 5.0, 30.0], np.float32),
                ior=np.array([[1.6, 1, 1, 1.55, 1], [1.61, 1, 1, 1.56, 1]], np.float32),
                semi_aperture=np.array([14.0, 14.0, 6.0, 13.0, 13.0], np.float32))
    check(tables(pkg, flat))
    # 15 interfaces, 8 wavelengths, every pair: the largest tables the ABI allows for
    n = 15
    big = dict(n=n, stop=7, radius=np.array([30.0 + 3 * k if k != 7 else 0.0 for k in range(n)], np.float32) *
               np.array([1 if k % 2 == 0 else -1 for k in range(n)], np.float32),
               thickness=np.full(n, 3.0, np.float32),
               ior=np.tile(np.array([1.5 if k % 2 == 0 else 1.0 for k in range(n)], np.float32), (8, 1)),
               semi_aperture=np.full(n, 10.0, np.float32))
    t = tables(pkg, big)
    assert len(t["paths"]) == 1 + 14 * 13 // 2
    check(t)


def test_invalid_selections_are_refused(pkg):
    lens = pkg.load_lens_file("dgauss11.lens")
    lib = pkg.load_library()
    info = np.zeros(8 + 4 * (MAX_PAIRS + 1), np.int32)
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))   # noqa: E731
    bad = np.array([[5, 7]], np.int32)      # 5 is the stop
    st = lib.lf_march_tables(11, 5, 3, f(lens["radius"]), f(lens["thickness"]), f(lens["ior"]),
                             f(lens["semi_aperture"]), bad.ctypes.data_as(C.POINTER(C.c_int)), 1, 1,
                             info.ctypes.data_as(C.POINTER(C.c_int)), None, C.c_size_t(0), None, C.c_size_t(0))
    assert st != 0
