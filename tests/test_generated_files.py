"""The committed generated files regenerate byte-identically from the reference tree (this container only: the GPU box has
no /root/reference, and nothing at test or bench time needs it there)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present (GPU box)")


def test_gen_tables_regenerates_the_committed_header(tmp_path):
    out = tmp_path / "rfc6716_tables.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_tables.py"), os.path.join(REF, "src"), str(out)],
                          stdout=subprocess.DEVNULL)
    assert out.read_bytes() == open(os.path.join(ROOT, "opus_amd", "csrc", "rfc6716_tables.h"), "rb").read()


def test_make_fixtures_regenerates_the_committed_kats(tmp_path):
    """tests/golden/kats.json = the known-answer vectors transcribed from the reference's own tests"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(ROOT, "tests", "golden", "make_fixtures.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    from pathlib import Path
    mf.REF = Path(REF)                       # (the script reads its tree from argv[1] when run by hand)
    assert mf.extract_kats() == json.load(open(os.path.join(ROOT, "tests", "golden", "kats.json")))
