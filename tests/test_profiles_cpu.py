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
