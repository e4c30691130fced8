"""Generate tests/golden/heads_reference.pt by RUNNING THE REFERENCE'S OWN CODE in this container:

  * `BlaserModel` from /root/reference/sonar/models/blaser/model.py (plain torch, imported by path);
  * `create_mutox_model` from /root/reference/sonar/models/mutox/factory.py + `MutoxClassifier` from
    model.py -- their only non-torch import is the `MutoxConfig` dataclass (config.py pulls in
    fairseq2, absent here), which is supplied as a stub module with the same single field.

The fixtures (small widths so they stay a few hundred KB) pin oracle/heads.py and, through it, the
HIP heads.  Run in the build container:  python tests/golden/make_golden_heads.py
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True   # importing the reference by path must not leave __pycache__ in /root/reference (read-only by contract)
from dataclasses import dataclass

import torch

REF = "/root/reference/sonar/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "heads_reference.pt")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.manual_seed(20240927)
    blaser = load("ref_blaser_model", f"{REF}/blaser/model.py")

    # MuTox: stub only the config dataclass, everything else is the reference's code
    for pkg in ("sonar", "sonar.models", "sonar.models.mutox"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    cfg_mod = types.ModuleType("sonar.models.mutox.config")

    @dataclass
    class MutoxConfig:
        input_size: int

    cfg_mod.MutoxConfig = MutoxConfig
    sys.modules["sonar.models.mutox.config"] = cfg_mod
    load("sonar.models.mutox.model", f"{REF}/mutox/model.py")
    mutox_factory = load("sonar.models.mutox.factory", f"{REF}/mutox/factory.py")

    out = {"blaser": [], "mutox": []}
    n, d = 9, 64
    for form, act, out_act, norm in (("COMET", "TANH", False, True), ("QE", "TANH", False, True),
                                     ("QE", "RELU", True, False), ("COMET", "TANH", True, True)):
        m = blaser.BlaserModel(embedding_dim=d, output_dim=1, hidden_dims=[128, 128], dropout=0.1, activation=act,
                               input_form=form, norm_emb=norm, output_act=out_act).eval()
        with torch.no_grad():   # fp16-representable weights: the fixture stores the matrices as halves
            for p in m.parameters():
                p.copy_((torch.randn_like(p) * (0.15 if p.dim() == 2 else 0.3)).half().float())
        src, mt, ref = (torch.randn(n, d) * s for s in (1.0, 0.7, 2.0))
        with torch.no_grad():
            y = m(src=src, mt=mt, ref=ref)
            feats = m.featurize_input(src=m._norm_vec(src), mt=m._norm_vec(mt), ref=m._norm_vec(ref))
        out["blaser"].append({"config": dict(embedding_dim=d, output_dim=1, hidden_dims=[128, 128], activation=act,
                                             input_form=form, norm_emb=norm, output_act=out_act),
                              "state_dict": {k: (v.half() if v.dim() == 2 else v.clone())
                                             for k, v in m.state_dict().items()},
                              "src": src, "mt": mt, "ref": ref, "features": feats, "out": y})
    for input_size in (256,):
        m = mutox_factory.create_mutox_model(MutoxConfig(input_size=input_size)).eval()
        with torch.no_grad():
            for p in m.parameters():
                p.copy_((torch.randn_like(p) * (0.08 if p.dim() == 2 else 0.3)).half().float())
        x = torch.randn(7, input_size) * 0.5
        with torch.no_grad():
            y, yp = m(x), m(x, output_prob=True)
        sd = {k: (v.half() if v.dim() == 2 else v.clone()) for k, v in m.state_dict().items()}
        out["mutox"].append({"input_size": input_size, "state_dict": sd, "x": x, "out": y, "prob": yp})
    torch.save(out, OUT)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
