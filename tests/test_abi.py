"""The C-ABI library builds, loads without a GPU and exports exactly the entry points include/sln_hip.h declares
(no compute calls here: this runs in the CPU-only container)."""
import os
import re
import subprocess

from conftest import ROOT, pkg

HEADER = os.path.join(ROOT, "include", "sln_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sln_[a-z0-9_]+)\s*\(", txt)))


def test_header_ctypes_table_and_library_agree():
    L = pkg("_lib")
    decl = _declared()
    assert len(decl) >= 30
    assert sorted(L.SIGNATURES) == decl, sorted(set(decl) ^ set(L.SIGNATURES))
    lib = L.lib()                                   # resolves every symbol or raises
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sln_[a-z0-9_]+)", nm))
    assert set(decl) <= exported, sorted(set(decl) - exported)
    assert lib.sln_version() >= 1 and lib.sln_build_arch() == b"gfx950"


def test_library_is_not_linked_against_a_second_hip_runtime():
    L = pkg("_lib")
    out = subprocess.run(["ldd", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" not in out, "must bind to the HIP runtime already in the process (torch's), see build.py"


def test_no_gpu_is_reported_not_hidden():
    import torch
    L = pkg("_lib")
    if not torch.cuda.is_available():
        assert L.lib().sln_device_ok() == -4            # SLN_E_NOGPU
        M = pkg("host.Sg2ScVAE_model")
        syn = pkg("host.synthetic")
        m = M.Sg2ScVAEModel(vocab=syn.default_vocab(), embedding_dim=16, gconv_num_layers=1, decoder_cat=True,
                            mlp_normalization='batch')
        b = syn.scene_graph_batch(2, 4, 6)
        try:
            m(b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"], None)
        except L.SlnError as e:
            assert "no CPU fallback" in str(e)
        else:
            raise AssertionError("the product path must fail loudly without the GPU")


def test_product_code_never_imports_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "3d_sln_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt and f.endswith(".py") and "import" in txt and "oracle." in txt:
                    bad.append(f)
    assert not bad, bad
