"""The committed measurement records must belong to the committed kernel sources (nothing enforced that
`roofline.traffic` was measured on the benchmarked build).  They are stamped with `csm_hf_amd.build.sources_sha256()`; this test fails
as soon as a kernel source changes without the PMC passes / in-step timelines being collected again (tools/collect_profiles_r05.sh,
tools/pmc_record.py), and bench.py marks the figures `stale` at run time by the same rule."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_records_match_the_kernel_sources():
    from csm_hf_amd.build import sources_sha256
    sha = sources_sha256()
    recs = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["records"]
    assert {(r["batch"], r["ctx"], r["weights"]) for r in recs} >= {(1, 512, "bf16"), (16, 512, "bf16")}
    for r in recs:
        assert r.get("src_sha256") == sha, ("profiles/hbm_traffic.json is stale for", r["batch"], r.get("commit"))
        # weights 9.0-9.7 GB per step; beyond 16 rows the fp32 KV cache of the rows (4.3 + 2.4 GB at 128 rows) and the planes' L2 misses add to it
        assert 0.9 * 9.0e9 < r["hbm_bytes_per_step"] < (1.6 if r["batch"] <= 16 else 3.2) * 9.7e9
    for b in (1, 16):
        lk = json.load(open(os.path.join(ROOT, "profiles", f"launch_kinds_b{b}.json")))
        assert lk["src_sha256"] == sha, f"profiles/launch_kinds_b{b}.json is stale"
        assert lk["batch"] == b and lk["launches_per_step"] == sum(k["launches"] for k in lk["kinds"])


def test_bench_marks_a_foreign_record_stale():
    sys.path.insert(0, ROOT)
    import bench
    rec = bench.traffic_record(1, 512, "bf16")
    dst = {}
    bench.attach_traffic(dst, rec, 9.02e9)
    assert dst["traffic_stale"] is False and 1.0 < dst["traffic_over_algorithmic"] < 1.1
    bad = dict(rec, src_sha256="0" * 64)
    bench.attach_traffic(dst, bad, 9.02e9)
    assert dst["traffic_stale"] is True
    assert bench.launch_kinds(1)["stale"] is False and bench.launch_kinds(7) is None


def test_default_kv_cache_dtype_outside_the_suite_pin():
    code = ("import torch; from csm_hf_amd import CSMConfig, CSMModel; m = CSMModel(CSMConfig.tiny()); "
            "assert CSMModel.DEFAULT_KV_DTYPE == 'auto' and m.kv_dtype == 'auto'")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
