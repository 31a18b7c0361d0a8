"""scripts/rank_nominal.py at reduced size: the legs bench.py reports as extra.cfg4_rank / extra.cfg5_rank (one rank's nominal
share of BASELINE.json's two 8-GPU configs on one GPU) -- corpus generated in chunks, assigned chunk by chunk to a trained
quantiser (vers_kmeans_assign_dev), STREAMED into a sharded handle (vers_ivf_upload_begin / _chunk_dev / _end), the rank's
partial results bitwise against the CPU restatement over the rank's sub-index; one rank's rows through
vers_ivf_build_sharded_dev with the properties the reference's build guarantees."""
import pytest

pytestmark = pytest.mark.gpu


def test_cfg4_rank_reduced():
    from scripts import rank_nominal as rn
    for rank, world in ((1, 4), (0, 1)):
        r = rn.cfg4_rank(rows=150_000, d=96, nlist=64, rank=rank, world=world, chunk=40_000, sample=50_000, iters=2, nprobe=8, B=64,
                         steps=3, check=12, log=lambda *a: None)
        assert r["check"]["gpu_matches_cpu_bitwise"] is True and r["check"]["queries"] >= 12
        up = r["upload"]
        assert up["stored_rows"] <= 150_000 and (world == 1 or up["stored_rows"] < 150_000 * 0.4)
        # no buffer of n_total rows: the library's peak stays near the stored rows' three copies (+ slack of small lists)
        assert up["library_bytes_peak"] < 6.0 * up["stored_rows"] * 96 * 4 + (64 << 20)
        assert r["search"]["step_ms"]["s1"] > 0 and r["search"]["step_ms"]["s3"] > 0


def test_cfg5_rank_reduced():
    from scripts import rank_nominal as rn
    r = rn.cfg5_rank(rows_total=160_000, world=4, d=64, k=512, iters=2, log=lambda *a: None)
    assert r["kept"] and r["properties"]["assignments_match_list_lengths"] and r["properties"]["self_retrieval_distance_exactly_zero"]
    assert r["iterations"] >= 1 and r["assign_passes"] >= 1   # (assign_passes counts MATRIX-CORE passes: a build this small may take the exact scan)
