"""CPU tier: `python -m oracle.gen_ref_fixtures --real-tf` -- the script that checks the committed reference fixtures against a REAL
TensorFlow wherever one exists.  Here (no TensorFlow in the image) it must say so and exit 2 without installing the shim or touching the fixtures."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_real_tf_mode_reports_a_missing_tensorflow_and_writes_nothing():
    gold = os.path.join(ROOT, "tests", "golden")
    before = {f: os.path.getmtime(os.path.join(gold, f)) for f in os.listdir(gold)}
    r = subprocess.run([sys.executable, "-m", "oracle.gen_ref_fixtures", "--real-tf", "vit_small"], cwd=ROOT, capture_output=True, text=True)
    try:
        import tensorflow  # noqa: F401
        have_tf = True
    except Exception:
        have_tf = False
    if have_tf:
        assert r.returncode in (0, 1), r.stdout + r.stderr
        assert "worst deviation of real TensorFlow" in r.stdout
    else:
        assert r.returncode == 2, r.stdout + r.stderr
        assert "tensorflow is not importable here" in r.stdout
    assert before == {f: os.path.getmtime(os.path.join(gold, f)) for f in os.listdir(gold)}
