"""Copies what scripts/profile_r06.sh left under gpurun_out/prof_r06 (merged back from the GPU box) into profiles/r06_*:
the summary, the PMC-derived traffic of the dominant kernel (read by bench.py), the other kernels' facts, the per-run rocprofv3
kernel_stats.csv.  Delete gpurun_out/prof_r06 before the run: the merge-back adds files, it does not remove old ones."""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", "prof_r06"), os.path.join(ROOT, "profiles")
facts = json.load(open(os.path.join(src, "facts.json")))
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, "r06_summary.txt"))
json.dump(facts.pop("cfg3"), open(os.path.join(dst, "r06_traffic.json"), "w"), indent=1)
json.dump(facts, open(os.path.join(dst, "r06_other_kernels.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "kmeans.json")):
    shutil.copy(os.path.join(src, "kmeans.json"), os.path.join(dst, "r06_kmeans.json"))
for tag in ("cfg3", "cfg3x", "cfg3_s3", "b1", "flat", "d1536", "edges", "kmeans", "shard2", "shard2_s3", "shard4", "shard4_s3", "shard8", "shard8_s3"):
    fs = glob.glob(os.path.join(src, tag, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if len(fs) != 1:
        sys.exit(f"{tag}: {len(fs)} kernel_stats.csv files (stale merge-back?)")
    shutil.copy(fs[0], os.path.join(dst, f"r06_{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "shard_all_ranks.json"), os.path.join(dst, "r06_shard_all_ranks.json"))
for f in ("cfg4_rank_nlist16384.json", "cfg4_rank_nlist4096.json", "cfg5_rank.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "r06_" + f))
print("profiles/r06_* updated")
