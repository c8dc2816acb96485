#!/usr/bin/env python3
"""tools/make_manifest.py <tag> <commit>: writes profiles/<tag>_manifest.json — the commit and the hash of the kernel sources the
counter files profiles/<tag>_* were measured on, and the sha256 of each of those files. bench.py reads it instead of asking git
(the driver's GPU box has no history) and marks replayed counters `stale` when the sources of the running build differ."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag, commit = sys.argv[1], sys.argv[2]
files = {}
for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
    if f.startswith(tag + "_") and not f.endswith("_manifest.json") and os.path.isfile(os.path.join(ROOT, "profiles", f)):
        files[f] = hashlib.sha256(open(os.path.join(ROOT, "profiles", f), "rb").read()).hexdigest()[:16]
out = {"tag": tag, "commit": commit, "csrc_sha": bench.csrc_sha(), "files": files,
       "note": "csrc_sha = sha256 over wgpu-path-tracing_amd/csrc/*.{hip,h} and include/*.h (names + contents), first 16 hex digits"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_manifest.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("tag", "commit", "csrc_sha")}), len(files), "files")
