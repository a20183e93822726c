#!/usr/bin/env python
"""Fold a PMC summary (tools/collect_pmc.sh -> pmc_hbm*.json) into profiles/hbm_traffic.json, stamped with the build it was
measured on: sha256 of the kernel sources (csm_hf_amd.build.sources_sha256; plus the library file's own hash and the commit of the tree).
bench.py compares the stamp with the library it runs and sets `roofline.traffic_stale` when they differ.

usage: python tools/pmc_record.py <pmc_hbm.json> <batch> <ctx> <weights> <source note> [opt=value ...]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csm_hf_amd.build import sources_sha256  # noqa: E402
src, batch, ctx, weights, note = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
opts = sorted(sys.argv[6:])
pm = json.load(open(src))
lib = os.path.join(ROOT, "csm-hf_amd", "libcsm_hip.so")
rec = {
    "batch": batch, "ctx": ctx, "weights": weights, "opts": opts,
    "hbm_bytes_per_step": pm["hbm_bytes_per_step"], "hbm_read_bytes_per_step": pm["hbm_read_bytes_per_step"],
    "src_sha256": sources_sha256(),
    "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
    "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
              + ("+dirty" if subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "csm-hf_amd", "include"],
                                            capture_output=True, text=True).stdout.strip() else ""),
    "source": note,
}
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
doc = json.load(open(path))
doc["records"] = [r for r in doc["records"] if not (r.get("batch") == batch and r.get("ctx") == ctx and r.get("weights", "bf16") == weights
                                                    and sorted(r.get("opts", [])) == opts)] + [rec]
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(rec, indent=1))
