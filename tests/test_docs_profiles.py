"""Every profile file that DESIGN.md, README.md, INTEGRATION.md, profiles/README.md or profiles/EXPERIMENTS.md names exists (the numbers in the documents
are only as good as the files they point to; the per-build file sets are renamed whenever a build is re-profiled)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = re.compile(r"`((?:profiles/)?r0\d_[A-Za-z0-9_*<>.\-]+\.(?:json|txt|csv|md))`")


def candidates(doc):
    text = open(os.path.join(ROOT, doc)).read()
    for m in NAME.finditer(text):
        name = m.group(1)
        if "<" in name:            # `r04_v45_<workload>_pmc_summary.json`: a pattern over the workloads
            name = re.sub(r"<[^>]*>", "*", name)
        yield name if name.startswith("profiles/") else "profiles/" + name


def test_documents_name_existing_profile_files():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "tools/README.md", "profiles/EXPERIMENTS.md"):
        for name in candidates(doc):
            if not glob.glob(os.path.join(ROOT, name)):
                missing.append((doc, name))
    assert not missing, missing


def test_current_pmc_profiles_belong_to_one_build():
    import json
    cur = json.load(open(os.path.join(ROOT, "profiles", "pmc_current.json")))
    ids = set()
    for workload, entry in cur.items():
        if isinstance(entry, dict) and "build_id" in entry:
            ids.add(entry["build_id"])
            assert os.path.exists(os.path.join(ROOT, entry["source"])), entry
    assert len(ids) == 1, ids


def test_committed_pmc_profile_is_of_the_library_that_is_built():
    """bench.py reports counter-derived roofline figures only when profiles/pmc_current.json was collected on the build it
    times (sart_build_id): a source edit without a re-profile would turn the bench line's roofline block into nulls."""
    import json
    from solaraxionraytracing_amd import _lib
    cur = json.load(open(os.path.join(ROOT, "profiles", "pmc_current.json")))
    assert cur["babyiaxo_xmm"]["build_id"] == _lib.build_id()
