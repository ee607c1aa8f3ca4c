"""bench.py's host-side pieces that need no GPU: the `cpu_baseline` leg (the oracle = the reference's torch CPU ops, timed on this box's cores;
a tiny sample here, the whole 32 Mb strand in the driver's run) and the projection helper of the strong-scaling note."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_fields_and_projection():
    b = _bench()
    r = b.cpu_baseline(0, sample_bp=1_600_000)
    assert r["kind"] == "port" and r["unit"] == "Mb/s" and r["cores"] >= 1 and r["value"] > 0 and r["extrapolated"] is True
    assert "EXTRAPOLATED x20" in r["sample"] and r["t_encoder_sample_s"] > 0 and r["t_decoders_s"] > 0
    p256 = b.projection(400.0, 388.0, 12.0, 8)
    p32 = b.projection(65.0, 50.0, 15.0, 8)
    assert p256["projected"] and p32["projected"] and 0.8 < p256["efficiency"] < 1.0 and 0.4 < p32["efficiency"] < 0.7
