"""Record hygiene (hygiene): every profiles/*.json is ONE parseable JSON document (a bench line's stderr
belongs in a .log / .err beside it), and the bench record of the current round names the sources it was taken from."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_profile_json_parses():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json")))
    assert len(files) > 50
    for f in files:
        with open(f) as fh:
            json.load(fh)


def test_the_random_search_draws_what_it_says():
    """tests/cull_fuzz.py without a device: its two synthetic design families are lenses (positive focal length, a stop
    inside, more than one glass), its masks have open and closed texels, and the stream of draws that found the 27 frames
    of tests/test_gpu_cull.py ONCE_LOST still starts the same way (the replay depends on it)."""
    import numpy as np
    import cull_fuzz
    for make, n in ((cull_fuzz.triplet, 7), (cull_fuzz.retrofocus, 9)):
        lens = make()
        assert lens["n"] == n and 0 < lens["stop"] < n - 1 and lens["radius"][lens["stop"]] == 0.0
        assert 30.0 < cull_fuzz.pkg.paraxial_efl(lens) < 90.0
        assert lens["ior"].shape == (3, n) and len({float(v) for v in lens["ior"][1] if v > 1.0}) >= 2
    cull_fuzz.rng = np.random.default_rng(424242)
    cull_fuzz.FAMILIES = 4
    first = [cull_fuzz.draw_lens()[:2] for _ in range(4)]
    assert all(name in ("dgauss11.lens", "dgauss11_8lambda.lens", "triplet") for name, _ in first)
    for kind in ("ring", "slit", "polygon", "speckle"):
        m = cull_fuzz.synthetic_mask(kind)
        assert m.shape == (256, 256) and 0.0 < float((m > 0).mean()) < 1.0
