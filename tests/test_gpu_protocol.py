"""Replay of the reference drivers' call protocol (tests/golden/G21_protocol.json: every call the reference's `genomepredict` and
`genomepredict_256Mb` make on a model, with the shapes AND strides of each argument) against this package's containers on the
MI355X: the arguments are built the way the reference builds them - the transposed view of a [1, L, 4] sequence, slices of the
encodings this package's own networks return, an expanded / flipped distance matrix, a crop of the previous level's prediction -
and must come out with the recorded strides; every call must return what the reference's code goes on to index."""
import json
import os
import re

import numpy as np
import pytest
import torch

from orca_amd import orca_models as M, synth

pytestmark = pytest.mark.gpu
G21 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "G21_protocol.json")))


def _same_view(t, rec):
    assert list(t.shape) == rec["shape"] and str(t.dtype) == "torch." + rec["dtype"], (list(t.shape), rec)
    assert list(t.stride()) == rec["stride"] and t.is_contiguous() == rec["contiguous"], (list(t.stride()), rec)


@pytest.mark.parametrize("fn", ["genomepredict", "genomepredict_256Mb"])
def test_reference_driver_call_sequence_on_the_device(cuda, fn):
    big = fn == "genomepredict_256Mb"
    model = (M.H1esc_256M if big else M.H1esc)(synthetic_seed=0)
    L = 256_000_000 if big else 32_000_000
    seq = torch.from_numpy(synth.synth_sequence(L, seed=3)).to(cuda)                 # [1, L, 4] as the reference holds it (:324-337)
    state = {}
    with torch.no_grad():
        for c in G21[fn]["calls"]:
            name, recs = c["call"], c["args"]
            if name == "net0":
                state.clear()
                args = [seq.transpose(1, 2)]                                          # `.transpose(1, 2).cuda()`
            elif name == "net1":
                args = [state["net0"]]
            elif name == "net":
                args = [state["net1"][-1] if big else state["net0"]]
                assert not big or isinstance(state["net1"], (list, tuple))
            else:
                lv = int(re.search(r"(\d+)", name).group(1)) if name != "denet_1_pt" else 1
                levels = [32, 64, 128, 256] if big else [1, 2, 4, 8, 16, 32]
                enc = state["net"][levels.index(lv)]                                  # the reference unpacks fine -> coarse (:333, :675)
                s0 = recs[0]["offset"]
                args = [enc[:, :, s0:s0 + 250]]
                if name != "denet_1_pt":
                    de = torch.log(torch.rand(1, 250, 250, device=cuda) + 0.5)[None].expand(1, -1, -1, -1)
                    args.append(torch.flip(de, [2, 3]) if big else de)
                    if len(recs) > 2:
                        i = recs[2]["offset"] // 251
                        args.append(state["pred"][:, :, i:i + 125, i:i + 125])
            assert len(args) == len(recs)
            for t, r in zip(args, recs):
                _same_view(t, r)
            sub = model.denets[int(re.search(r"(\d+)", name).group(1))] if name.startswith("denets") else getattr(model, name)
            out = sub.forward(*args) if name.startswith("den") else sub(*args)        # the reference calls `.forward` on the decoders
            outs = list(out) if isinstance(out, (list, tuple)) else [out]
            assert [list(o.shape) for o in outs] == c["returns"] and all(o.is_cuda and o.dtype == torch.float32 for o in outs), (name, [o.shape for o in outs])
            assert isinstance(out, (list, tuple)) == c["returns_list"], name
            if name == "denet_1_pt":
                state["pred"] = state["pred"] + out                                   # `... + model.denet_1_pt.forward(...)` (:362)
            elif name.startswith("denets"):
                state["pred"] = out
            else:
                state[name] = out
            assert all(bool(torch.isfinite(o).all()) for o in outs), name
