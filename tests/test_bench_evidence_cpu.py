"""bench.py's roofline records cite PMC evidence of ONE round (VERDICT r4 item 3 / ADVICE r4 #5): the files exist, carry the source hash of the
tree they were collected on, and bench.py marks them stale -- instead of presenting them as current -- when the running tree hashes differently."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_source_hash_is_deterministic_and_ignores_built_artifacts():
    h = bench.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h)
    assert h == bench.source_hash()


def test_evidence_files_of_the_reported_round_are_stamped():
    r = bench.EVIDENCE_ROUND
    for name in ("pmc_traffic.json", "binding.json"):
        d = bench._evidence(name)
        assert d is not None, f"profiles/{r}_{name} missing"
        assert re.fullmatch(r"[0-9a-f]{16}", d["_meta"]["source_hash"]), d["_meta"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    other = set(re.findall(r"profiles/(r\d\d)_", src)) - {r}
    assert not other, f"bench.py reads or cites evidence of other rounds: {other}"


def test_binding_record_says_where_it_comes_from_and_whether_it_is_current():
    b = bench.binding_metric("gs_env_shade_fwd")
    assert b and b["source"].startswith(f"profiles/{bench.EVIDENCE_ROUND}_")
    assert b["collected_at_source_hash"] == bench._evidence("binding.json")["_meta"]["source_hash"]
    if b["collected_at_source_hash"] != bench.source_hash():
        assert b.get("stale_sources_now") == bench.source_hash()
    else:
        assert "stale_sources_now" not in b


def test_recorded_gpu_suite_finishes_well_inside_the_drivers_limit():
    """profiles/r06_gpu_test_durations.txt = `pytest tests -m gpu --durations=0` + smoke() of the round's tree on one MI355X lease (the driver gives the
    suite 1200 s; round 5 ran into that wall).  The recorded wall time must stay under 600 s, no single test above 60 s, nothing failed."""
    path = os.path.join(ROOT, "profiles", "r06_gpu_test_durations.txt")
    assert os.path.isfile(path), "profiles/r06_gpu_test_durations.txt missing: run tools/collect_r06.sh on the GPU box"
    text = open(path).read()
    m = re.search(r"(\d+) passed(?:, (\d+) skipped)?.* in ([0-9.]+)s", text)
    assert m, "no pytest summary line in the record"
    assert " failed" not in text.split("slowest durations")[-1].splitlines()[-8:].__str__() and "rc=0" in text
    assert float(m.group(3)) <= 600.0, f"the recorded GPU suite took {m.group(3)} s"
    calls = [float(x) for x in re.findall(r"^([0-9.]+)s call ", text, flags=re.M)]
    assert calls and max(calls) <= 60.0, f"slowest test {max(calls)} s"
    s = re.search(r"smoke ok.*\n.*smoke_seconds=([0-9.]+)", text)
    assert s and float(s.group(1)) <= 120.0, "smoke() not recorded or slower than 120 s"
