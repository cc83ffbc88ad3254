"""bench.py's launcher logic that can be checked without a GPU: a multi-GPU request on a box with fewer devices must fail
loudly instead of running one rank and reporting it as the requested job."""
import os
import subprocess
import sys

import torch

from conftest import ROOT


def test_multi_gpu_request_without_devices_fails_loudly():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return                                            # a real multi-GPU box: the request is legitimate there
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout


def test_world_size_must_match_the_request():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_recorded_bench_line_carries_the_contract_keys():
    """profiles/r03_bench.json is the line `python bench.py` printed on the MI355X for the committed code: the keys the driver
    and the judge read must all be there (the default run fills every leg, each with its roofline and cpu_baseline)."""
    import json
    path = os.path.join(ROOT, "profiles", "r03_bench.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 64 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-3 * d["value"]
    for leg in (d, d["render"], d["spade"]):
        r, c = leg["roofline"], leg["cpu_baseline"]
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["bound"] in ("hbm", "mfma")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert d["c1"]["cores"] >= 1 and d["c1"]["cpu_ms_per_iter"] > 0
    assert all(v is not None for v in d["parity"].values()) and d["render"]["parity"]["face_index_pixels_differing"] == 0
    # round 3: a number for the algorithmic bytes, every GEMM family with its own roofline, the unchanged call sequences
    assert isinstance(d["roofline"]["algorithmic_bytes_per_launch"], int) and d["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert {"gemm_nt", "gemm_tn"} <= set(d["roofline_kernels"])
    for fam in d["roofline_kernels"].values():
        assert abs(fam["frac"] - fam["achieved"] / fam["peak"]) < 2e-3 and fam["launches_per_step"] > 0
    assert d["vae_dropin"]["fused_adam"]["ratio_to_fused_step"] <= 1.5 and d["vae_dropin"]["torch_adam"]["ms_per_step"] > 0
    assert d["render"]["render_33pass"]["maps_reused"]["ms_per_render"] > 0 and d["spade"]["spade_50x1"]["images_per_s"] > 0
    for k in ("scene_compose_kernel", "scene_bwd_plane_sums_kernel"):
        assert d["render"]["roofline_kernels"][k]["bound"] == "hbm" and d["render"]["roofline_kernels"][k]["frac"] <= 1.0


def test_round6_bench_line_ends_with_the_summary_of_every_config():
    """profiles/r06_bench.json: the LAST key of the line is `summary` (what the tail of a log shows): both halves of the metric - c2
    graphs/s and c3 renders/s, mean and median -, the other legs, every roofline fraction and every in-run parity figure; the render
    figure is the wall-clock mean and agrees with the median; roofline traffic and algorithmic bytes come from one launch set; the
    refinement leg carries its parity against the reference-executed loop."""
    import json
    line = open(os.path.join(ROOT, "profiles", "r06_bench.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    assert list(d.keys())[-1] == "summary" and len(json.dumps(d["summary"])) <= 700
    s = d["summary"]
    for k in ("c2_graphs_per_s", "c2_frac_mfma_step", "c3_renders_per_s", "c3_renders_per_s_median", "c3_frac_hbm", "c4_images_per_s",
              "c4_frac_mfma", "refine16_ms", "refine64_ms", "sampling_layouts_per_s", "c2_err", "c3_px_diff", "c4_err", "refine_err"):
        assert s.get(k) is not None, k
    assert s["c2_graphs_per_s"] == d["value"] and s["c3_renders_per_s"] == d["render"]["renders_per_s"]
    assert d["render"]["mean_over_median"] <= 1.05
    assert max(s["c2_err"], s["c4_err"], s["refine_err"]) <= 1e-4 and s["c3_px_diff"] == 0
    r = d["roofline"]
    assert abs(r["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 2e-3, "one launch set for numerator and denominator"
    rs = d["roofline_step"]
    assert rs["traffic"] > 0 and abs(rs["traffic_over_survey_bytes"] - rs["traffic"] / rs["survey_bytes_per_step"]) < 2e-3
    assert d["refine"]["parity"]["loss"] <= 1e-4 and "finetune_VAE" in d["refine"]["parity"]["against"]


import pytest       # noqa: E402


@pytest.mark.gpu
def test_bench_data_parallel_path_end_to_end_on_one_gpu():
    """bench.py's own launcher -> RCCL -> JSON line, end to end, on the one GPU a test box has: SLN_BENCH_FORCE_DP=1 takes the
    data-parallel route (graph without Adam, all-reduce of [gradients | guard] over RCCL with a world of one, fused Adam) that
    the 8-GPU run takes - the only part of that run this box cannot exercise is a second rank."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SLN_BENCH_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu",
                        "--no-render", "--no-spade", "--no-graph-build", "--no-refine", "--no-sampling", "--no-dropin", "--large-batches=", "--prof-steps", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp1"
    assert d["config"]["collective"] and "all-reduce" in d["config"]["collective"]
    assert d["config"]["allreduce_us_standalone"] is not None and d["config"]["allreduce_us_standalone"] > 0
    import math
    assert math.isfinite(d["config"]["final_total_loss"]) and d["value"] > 0 and "roofline" in d
