"""bench.py's own bookkeeping on the CPU (no GPU, no kernels): the rank launcher and the --gpus contract, the per-kernel roofline
table (dominant kernel, CU-limited exclusion, flops of sparse launches), the staleness rule of `roofline.traffic`."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _run(args, env=None, timeout=240):
    e = dict(os.environ, RD_BENCH_DRY_RUN="1")
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None); e.pop("MASTER_PORT", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout, text=True)


def test_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no torchrun) must run two ranks -- tools/scripts/dist_train.sh:10 of the reference starts them
    itself too -- and rank 0 alone prints the line (dry run: gloo rendezvous + barrier + max-over-ranks, no GPU)."""
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "8"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["global_batch"] == 16
    assert abs(d["max_over_ranks_s"] - 0.002) < 1e-9          # rank 1 reported 0.002 s, rank 0 0.001 s: the MAX is what counts


def test_gpus_flag_must_match_world_size():
    r = _run(["--gpus", "8"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "1"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    with pytest.raises(SystemExit):
        bench.check_world(2, 4)
    bench.check_world(4, 4)


def test_a_failing_rank_fails_the_launcher():
    # without the dry-run switch the ranks stop at "needs an MI355X" here (no GPU in this container): the launcher must relay that
    r = _run(["--gpus", "2"], env={"RD_BENCH_DRY_RUN": ""})
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run the real benchmark")
    assert r.returncode != 0 and "needs an MI355X" in r.stderr


class _Ev:
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return other.t - self.t


def _launch(t0, ms, flops, tag, rows=1000, cin=256, cout=256, taps=9, pairs=None):
    return (_Ev(t0), _Ev(t0 + ms), pairs, flops, (rows, cin, cout, taps, 1, tag))


def test_roofline_table_names_the_dominant_kernel_and_excludes_cu_limited_ones():
    table = [_launch(0, 2.0, 4e9, "d3_128"), _launch(2, 1.0, 1e9, "64_dense"), _launch(3, 5.0, 9e9, "wgrad_d3"), _launch(8, 1.5, 2e9, "d3_128")]
    timed = [_launch(0, 2.0, 4e9, "d3_128"), _launch(5, 1.0, 4e9, "d3_128")]
    roof, dom, key, dom_ms, all_ms, limited, fl = bench.mfma_roofline(timed, table, 1, 1, "bf16x3", {"wgrad_d3"})
    assert dom == "d3_128" and "k_conv_d3_b3<8, 16, 128" in key          # wgrad_d3 has the largest time but is CU-limited
    assert limited == {"wgrad_d3": 5.0}
    assert roof["avg_launch_ms"] == 1.5 and roof["peak"] == bench.PEAK_BF16_MFMA_TFLOPS
    assert abs(roof["achieved"] - 4e9 / 1.5e-3 / 1e12) < 1e-3
    assert abs(roof["frac"] - roof["achieved"] / 2500.0) < 1e-4 and abs(roof["mfma_issue_frac"] - 3 * roof["frac"]) < 2e-4
    assert roof["ms_per_step_by_kernel"]["d3_128"] == 3.5 and roof["launches_per_step"] == 2
    # without the exclusion the same table names the weight-gradient kernel; exact fp32 prices against the fp32 matrix peak
    roof2, dom2, *_ = bench.mfma_roofline(timed, table, 1, 1, "f32", ())
    assert dom2 == "wgrad_d3" and roof2["peak"] == bench.PEAK_F32_MFMA_TFLOPS and roof2["mfma_issue_frac"] == roof2["frac"]


def test_roofline_counts_sparse_flops_from_the_pair_count():
    pairs = torch.tensor(1000)
    timed = [_launch(0, 1.0, 2.0 * 64 * 128, "64x128_table", pairs=pairs)]
    roof, dom, *_ = bench.mfma_roofline(timed, timed, 1, 1, "bf16x3", ())
    assert roof["algorithmic_flops_per_launch"] == 1000 * 2 * 64 * 128


def test_traffic_is_null_when_the_profile_is_from_other_sources(tmp_path, monkeypatch):
    from radardistill_amd import native
    prof = tmp_path / "profiles"
    prof.mkdir()
    key = "k_conv_d3_b3<8, 16, 128"
    body = {"kernels": {f"void {key}, true>(ConvArgs, int)": {"hbm_bytes_per_launch_corrected": 123}}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    json.dump(dict(body, csrc_sha="not-the-current-sources"), open(prof / "round3_pmc_hbm_traffic_bf16x3.json", "w"))
    val, src = bench.pmc_traffic("bf16x3", key)
    assert val is None and "stale" in src
    json.dump(dict(body, csrc_sha=native.csrc_sha()), open(prof / "round3_pmc_hbm_traffic_bf16x3.json", "w"))
    val, src = bench.pmc_traffic("bf16x3", key)
    assert val == 123 and src.endswith("round3_pmc_hbm_traffic_bf16x3.json")
    assert bench.pmc_traffic("f32", key)[0] is None


def test_every_kernel_tag_has_a_name_in_both_modes():
    for b3 in (True, False):
        names = bench.kernel_names(b3)
        for tag in ("d3_128", "d3_16x64", "d3_64", 128, 64, "64_dense", "64x128_table", "wgrad_d3", "wgrad_b3_128", "wgrad_f32_128"):
            assert tag in names and len(names[tag]) == 2
