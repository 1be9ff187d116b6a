"""docs/K1_MODEL.md: the K1 time model (max of the traffic-only and the arithmetic-only kernel, plus the part of the smaller
side that does not hide) must keep reproducing the committed occupancy sweep of the product, both formats, within 5 %."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_k1_model_reproduces_the_measured_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k1_model.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "worst |error|" in r.stdout
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import k1_model as m
    a, b = m.load(box="== box A =="), m.load()
    for fmt in ("cf32", "u8"):                                    # another box's products from its own arithmetic side and box B's traffic side
        for k in sorted({key[1] for key in a}):
            p = m.model(b[("abl1s", k, fmt)], a[("abl6", k, fmt)])
            assert abs(p / a[("cur", k, fmt)] - 1.0) < 0.05, (fmt, k, p, a[("cur", k, fmt)])
