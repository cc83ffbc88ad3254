"""INTEGRATION.md section 2, executed: the reference's OWN factory (build_dataset_model.build_model) constructs the HIP-backed model
when ``models.graph`` / ``models.Sg2ScVAE_model`` are aliased, with the reference model's parameter names and shapes.
Runs only where the reference tree is mounted (the build container); read-only, no bytecode is written."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"

SCRIPT = r'''
import importlib, json, sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, %(ref)r)
sys.path.insert(0, %(root)r)
import models.graph as ref_graph                       # the reference's own modules, kept under other names for the comparison
import models.Sg2ScVAE_model as ref_vae
_hip = lambda m: importlib.import_module("3d_sln_amd.host." + m)
sys.modules["models.graph"] = _hip("graph")
sys.modules["models.Sg2ScVAE_model"] = _hip("Sg2ScVAE_model")
sys.modules["data.suncg_dataset"] = _hip("suncg_dataset")
sys.modules.pop("build_dataset_model", None)
import build_dataset_model as B                         # the reference's factory, now importing the aliased modules
vocab = _hip("synthetic").default_vocab()
args = types.SimpleNamespace(batch_size=64, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode="feedforward", gconv_num_layers=5,
                             mlp_normalization="batch", vec_noise_dim=0, layout_noise_dim=32, use_AE=False, multigpu=False)
model, kwargs = B.build_model(args, vocab)
assert type(model).__module__.startswith("3d_sln_amd.host"), type(model)
ref = ref_vae.Sg2ScVAEModel(**kwargs)
a = {k: tuple(v.shape) for k, v in model.state_dict().items()}
b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
assert a == b, sorted(set(a.items()) ^ set(b.items()))[:10]
assert [n for n, _ in model.named_parameters()] == [n for n, _ in ref.named_parameters()]       # optimizer state is positional
model.load_state_dict(ref.state_dict())                 # a reference checkpoint loads unchanged
import torch
torch.optim.Adam(model.parameters(), lr=1e-4)
# C. the generator, constructed as testing/test_SPADE_shade.py:9 does, small width to keep the test light
import models.SPADE_related as ref_spade
hs = _hip("SPADE_related")
cargs = (41, 3, 16, 8, "spectralspadelayer3x3", 64, "normal")
g_ref, g_hip = ref_spade.SPADEGenerator4(*cargs), hs.SPADEGenerator4(*cargs)
ga = {k: tuple(v.shape) for k, v in g_hip.state_dict().items()}
gb = {k: tuple(v.shape) for k, v in g_ref.state_dict().items()}
assert ga == gb, sorted(set(ga.items()) ^ set(gb.items()))[:10]
g_hip.load_state_dict(g_ref.state_dict())
print(json.dumps({"keys": len(a), "params": sum(p.numel() for p in model.parameters()), "spade_keys": len(ga)}))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_reference_factory_builds_the_aliased_model():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(ref=REF, root=ROOT)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["params"] == 3879790 and out["keys"] > 200
    assert out["spade_keys"] == 230                                # the 230-key checkpoint layout
