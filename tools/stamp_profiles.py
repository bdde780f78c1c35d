"""profiles/<round>_source.json: which sources (content hash of katana.jl_amd/csrc/*.hip|*.hpp, as bench.py computes it) and which
commit the round's committed rocprofv3 summaries were made from.  Run HERE (the GPU box has no .git) right after copying the
summaries of tools/profile_round.sh from gpurun_out/ into profiles/, with the tree the profile was made from checked out.

    python tools/stamp_profiles.py r04
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rnd = sys.argv[1]
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "katana.jl_amd/csrc"], capture_output=True, text=True).stdout.strip())
out = {"round": rnd, "csrc_sha16": bench.csrc_sha16(), "git_head": head + ("+uncommitted csrc changes" if dirty else ""),
       "note": "bench.py quotes the rocprofv3 figures of profiles/%s_* only while this hash equals the hash of the sources it runs from" % rnd}
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_source.json"), "w"), indent=1)
print(out)
