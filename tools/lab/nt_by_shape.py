"""Per-shape durations of the NT launches of ONE eager training step (GPU box).
   SLN_NT_LOG=1 makes the launcher print a line per launch; this script runs itself under rocprofv3 --kernel-trace, then joins the
   NT dispatches of the trace - in order - with those lines.      python tools/lab/nt_by_shape.py [graphs]"""
import csv, glob, importlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(B):
    import torch
    M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model"); syn = importlib.import_module("3d_sln_amd.host.synthetic")
    torch.manual_seed(42)
    model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=B, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                            gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
    model.validate_inputs = False
    b = syn.scene_graph_batch(B, 32, 64, seed=77, device="cuda")
    batch = (b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"])
    for i in range(3):
        if i == 2:
            sys.stderr.write("NTLOG BEGIN\n"); sys.stderr.flush()
        model.train_step(*batch, kl_weight=0.1, lr=1e-4, use_graph=False)
        torch.cuda.synchronize()
    sys.stderr.write("NTLOG END\n")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if os.environ.get("NT_CHILD"):
        return child(B)
    out = "/tmp/nt_by_shape"
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, NT_CHILD="1", SLN_NT_LOG="1", TMPDIR="/tmp")
    r = subprocess.run(["timeout", "600", "rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
                        os.path.abspath(__file__), str(B)], env=env, cwd="/tmp", capture_output=True, text=True)
    log = [l for l in r.stderr.split("\n") if l.startswith("NTLOG")]
    n_before = log.index("NTLOG BEGIN"); shapes_all = [l for l in log if "M=" in l]
    n_skip = sum("M=" in l for l in log[:n_before])
    last = [l for l in log[n_before + 1:] if "M=" in l]
    f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    nt = [x for x in rows if re.search(r"gemm_nt(_small|16)?_kernel", x["Kernel_Name"])]
    assert len(nt) == len(shapes_all), (len(nt), len(shapes_all))
    nt = nt[n_skip:n_skip + len(last)]
    agg = {}
    for l, x in zip(last, nt):
        us = (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3
        m = dict(kv.split("=") for kv in l.split()[1:])
        kn = re.sub(r"void \(anonymous namespace\)::|\(GemmNTArgs.*", "", x["Kernel_Name"])
        key = (int(m["M"]), int(m["N"]), int(m["K"]), m["amode"], m["epi"], m["nseg"], kn)
        agg.setdefault(key, []).append(us)
    tot = 0.0
    print("%8s %5s %5s  am ep sg  %-42s %3s %9s %7s %6s" % ("M", "N", "K", "kernel", "n", "avg us", "TF/s", "frac"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v); fl = 2.0 * k[0] * k[1] * k[2]; tot += sum(v)
        print("%8d %5d %5d  %s  %s  %s  %-42s %3d %9.1f %7.1f %6.3f" % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], len(v), avg, fl / avg / 1e6, fl / avg / 1e6 / 157.3))
    print("NT launches of one step: %d, %.3f ms" % (len(last), tot / 1e3))


if __name__ == "__main__":
    main()
