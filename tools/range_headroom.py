"""fp16-range evidence WITHOUT the published checkpoints (orca_models.py:53-123 load files that are absent offline).

The default arithmetic ("f16x2": every fp32 operand as two fp16 parts) has the fp16 exponent range: an activation >= 65 504 cannot be
split, the device raises the context's range flag and the module redoes its forward in the range-safe arithmetic (bf16x3 for the
Encoders, exact fp32 for the Decoders).  With synthetic weights the question "how far is that?" has two answers,
both produced here on the MI355X:

 * `headroom(model, codes)` - the LAYER-BY-LAYER exact-fp32 walk of tools/validate_checkpoints.py over the networks of one container:
   max |activation| per network and 65 504 / that (the headroom at gain 1).  bench.py puts it in its line as `fp16_headroom`.
 * `trip_sweep()` - the conv gain of the synthetic weights (orca_amd.synth.synth_state_dict(relu_gain=g): every conv weight x g, so the
   activations grow geometrically with depth) raised until the guard FIRES, per network: the gain at which it first fires, the largest
   activation of the fp32 walk there, and - on both sides of the trip point - the forward's agreement with the exact-fp32 mode (below the
   trip: the f16x2 result; at and above: the result of the automatic retry).  tests/test_gpu_nets.py asserts it.

What a tripped guard costs: the forward is run twice, the second time in the slower arithmetic - a whole 32 Mb step 155.6 instead of
64.3 ms (2.4 x; `bf16x3` in the bench line), never a wrong result.

    python tools/range_headroom.py            # JSON on stdout
"""
import json
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import validate_checkpoints as V
from orca_amd import orca_modules as pm, synth

F16_MAX = 65504.0
GAINS = (1.0, 1.3, 1.6, 2.0, 2.5, 3.2, 4.0, 5.0, 6.4, 8.0, 10.0, 13.0, 16.0, 20.0, 26.0, 32.0)
GAINS_DECODER = (1.0, 1.1, 1.2, 1.3, 1.4, 1.5, 1.6, 1.8, 2.0, 2.5, 3.2)     # 116 convs deep: the activations grow ~3 000 x between gain 1 and 1.6


def _module(cls, seed, gain, **kw):
    m = getattr(pm, cls)(**kw)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=seed, relu_gain=gain)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval()


def _lut(dev):
    return torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [.25, .25, .25, .25]], dtype=torch.float32, device=dev)


def headroom(model, codes_dev, sample_bp=1_600_000):
    """{network: {"max_abs_activation", "at", "headroom"}} of one 32 Mb container (H1esc / Hff) on ``codes_dev`` [1, 32 000 000] uint8:
    the Encoder on the first ``sample_bp`` bases, Encoder2 on the whole window's encoding, every Decoder level on its own inputs."""
    from orca_amd import orca_predict as P
    dev = codes_dev.device
    out = {}

    def put(name, rows):
        k, v = max(rows, key=lambda t: t[1])
        out[name] = {"max_abs_activation": round(v, 4), "at": k, "headroom": round(F16_MAX / max(v, 1e-30), 1)}
    with torch.no_grad():
        n = sample_bp - sample_bp % 4000
        x = _lut(dev)[codes_dev[0, :n].long()].t()[None].contiguous()
        put("Encoder", V.encoder_walk(model.net0, x)[0])
        del x
        enc0 = model.net0.forward_codes(codes_dev)
        r2, encs = V.unet_walk(model.net, enc0, 5, "Encoder2")
        put("Encoder2", r2)
        rows, ypred = [], None
        for lv in (32, 16, 8, 4, 2, 1):
            xs = encs[{1: 0, 2: 1, 4: 2, 8: 3, 16: 4, 32: 5}[lv]][:, :, :250].contiguous()
            de = P._cached_log_background(model, lv, True)
            rows += V.decoder_walk(model.denets[lv], xs, de.expand(1, -1, -1, -1).contiguous(), ypred, f"Decoder {lv}Mb")[0]
            ypred = model.denets[lv](xs, de.expand(1, -1, -1, -1), ypred)[:, :, :125, :125].contiguous()
        put("Decoders", rows)
        put("Decoder_1m", V.decoder_walk(model.denet_1_pt, encs[0][:, :, :250].contiguous(), None, None, "Decoder_1m")[0])
    out["min_headroom"] = min(v["headroom"] for v in out.values())
    out["what"] = ("65 504 / the largest |activation| of a layer-by-layer exact-fp32 walk (tools/validate_checkpoints.py) of this line's own model "
                   f"(synthetic weights; Encoder on {n} bases, the rest on the whole window) - how far the default f16x2 arithmetic is from its range guard")
    return out


def _forwards(kind, dev, seed):
    """(make(gain) -> module, inputs, run(module) -> output, walk(module) -> rows) of one network kind on small synthetic inputs."""
    rs = np.random.RandomState(100 + seed)
    if kind == "Encoder":
        x = torch.from_numpy(synth.synth_sequence(4000 * 40, seed=30 + seed, n_frac=0.01)).transpose(1, 2).contiguous().to(dev)
        return (lambda g: _module("Encoder", seed, g), lambda m: m(x), lambda m: V.encoder_walk(m, x)[0])
    if kind == "Encoder2":
        e = torch.from_numpy((rs.rand(1, 128, 2048) * 0.8).astype(np.float32)).to(dev)
        return (lambda g: _module("Encoder2", seed, g), lambda m: torch.cat([t.reshape(1, -1) for t in m(e)], 1), lambda m: V.unet_walk(m, e, 5, "Encoder2")[0])
    if kind == "Decoder":
        nm, _ = synth.synth_normmats_32m()
        e = torch.from_numpy((rs.rand(1, 128, 250) * 0.5).astype(np.float32)).to(dev)
        de = torch.log(torch.from_numpy(nm[4][None, None].astype(np.float32))).to(dev)
        y = torch.from_numpy(rs.randn(1, 1, 125, 125).astype(np.float32)).to(dev)
        return (lambda g: _module("Decoder", seed, g, upsample_mode="bilinear"), lambda m: m(e, de, y), lambda m: V.decoder_walk(m, e, de, y, "Decoder")[0])
    raise ValueError(kind)


def trip_sweep(kinds=("Encoder", "Encoder2", "Decoder"), seed=3, gains=None, dev=None):
    """Per network kind: rows [{gain, max_abs_activation, guard_fired, rel_err_vs_f32}] up to and including the first gain at which the
    range guard fires, and `trip_gain` (None: never within ``gains``)."""
    dev = dev or torch.device("cuda:0")
    rep = {}
    for kind in kinds:
        make, run, walk = _forwards(kind, dev, seed)
        rows, trip = [], None
        for g in (gains or (GAINS_DECODER if kind == "Decoder" else GAINS)):
            m = make(g).to(dev)
            with torch.no_grad():
                mx = max(v for _, v in walk(m))
                if not np.isfinite(mx):
                    break
                m.precision = "f32"
                ref = run(m).float().cpu().numpy()
                m.precision = "f16x2"
                with warnings.catch_warnings(record=True) as wl:
                    warnings.simplefilter("always")
                    got = run(m).float().cpu().numpy()
                fired = any("fp16 range" in str(w.message) for w in wl)
            scale = max(1.0, float(np.abs(ref).max()))
            rows.append({"gain": g, "max_abs_activation": round(float(mx), 2), "guard_fired": fired,
                         "rel_err_vs_f32": float(np.abs(got - ref).max()) / scale, "finite": bool(np.isfinite(got).all())})
            if fired:
                trip = g
                break
        rep[kind] = {"trip_gain": trip, "rows": rows,
                     "headroom_at_gain_1": round(F16_MAX / max(rows[0]["max_abs_activation"], 1e-30), 1) if rows else None}
    rep["what"] = ("synthetic conv weights x gain (orca_amd.synth.synth_state_dict(relu_gain=gain)) until the device range guard of the f16x2 arithmetic fires; "
                   "rel_err_vs_f32 = max-abs difference to the exact-fp32 mode / max(1, output range): the f16x2 result below the trip gain, the "
                   "automatic range-safe retry's result at it")
    return rep


if __name__ == "__main__":
    from orca_amd import orca_models as M
    dev = torch.device("cuda:0")
    model = M.H1esc(synthetic_seed=0)
    model.cuda()
    codes = torch.from_numpy(synth.synth_base_codes(32_000_000, seed=1))[None].to(dev)
    print(json.dumps({"headroom": headroom(model, codes), "trip_sweep": trip_sweep(dev=dev)}))
