"""The model protocol the reference's own drivers use (tests/golden/G21_protocol.json, recorded from the reference's `genomepredict` /
`genomepredict_256Mb` by tools/make_protocol_golden.py): every attribute they touch exists on this package's containers, is an
nn.Module, and its forward binds the recorded positional arguments - checked without a GPU."""
import inspect
import json
import os
import re

import torch.nn as nn

from orca_amd import orca_models as M, orca_predict as P

G21 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "G21_protocol.json")))


def _resolve(model, call):
    m = re.fullmatch(r"denets\[(\d+)\]", call)
    return model.denets[int(m.group(1))] if m else getattr(model, call)


def test_containers_expose_the_protocol_of_the_reference_drivers():
    for fn, model in (("genomepredict", M.H1esc(synthetic_seed=0)), ("genomepredict_256Mb", M.H1esc_256M(synthetic_seed=0))):
        assert isinstance(model, nn.Module)          # orca_predict.py:301-313: anything else is treated as a key of model_dict_global
        seen = set()
        for c in G21[fn]["calls"]:
            sub = _resolve(model, c["call"])
            assert isinstance(sub, nn.Module), c["call"]
            inspect.signature(sub.forward).bind(*c["args"], **{k: None for k in c["kwargs"]})   # raises TypeError if it cannot be called that way
            seen.add(c["call"])
        if fn == "genomepredict":
            assert seen == {"net0", "net", "denet_1_pt"} | {f"denets[{lv}]" for lv in (1, 2, 4, 8, 16, 32)}
            for lv in (1, 2, 4, 8, 16, 32):         # :350, :441
                assert model.normmats[lv].shape[-2:] == (250, 250) and float(model.epss[lv]) > 0
        else:
            assert seen == {"net0", "net1", "net"} | {f"denets[{lv}]" for lv in (32, 64, 128, 256)}


def test_driver_signatures_match_the_reference():
    """orca_predict.py:231-233, :543-556."""
    s = inspect.signature(P.genomepredict)
    assert list(s.parameters) == ["sequence", "mchr", "mpos", "wpos", "models", "targets", "annotation", "use_cuda", "nan_thresh"]
    assert (s.parameters["mpos"].default, s.parameters["wpos"].default, s.parameters["use_cuda"].default, s.parameters["nan_thresh"].default) == (-1, -1, True, 1)
    s = inspect.signature(P.genomepredict_256Mb)
    assert list(s.parameters) == ["sequence", "mchr", "normmats", "chrlen", "mpos", "wpos", "models", "targets", "annotation", "padding_chr", "use_cuda", "nan_thresh"]
