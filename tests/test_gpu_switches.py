"""Every build-time alternative the engine still carries behind an environment switch (A/B paths of earlier rounds: the
host-side chain selection, the byte-wide trace-back spill, the library's anchor sort, seeding and sorting as two kernels, compacted
minimizers, untagged two-piece cells, the 64-bit sketch, the int32 wide classes, serial class launches, chunked packed
launches, trace-back lane limits, synchronous result DMA, the table filter in the vote presets' lookups, which queries vote with 16-bit counters) must produce the SAME bits as the default path: each switch runs
the randomised HIP-vs-oracle parity (tests/fuzz_parity.py: every stage compared) in a process of its own, because the
switches are read once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {"TELR_TB8": "1"}, {"TELR_SORT64": "1"}, {"TELR_MZ_COMPACT": "1"},
    {"TELR_NO_TAG8": "1"}, {"TELR_SKETCH64": "1"}, {"TELR_NO_PKW": "1"}, {"TELR_SERIAL": "1"}, {"TELR_PK_CHUNKS": "3"},
    {"TELR_TB_SPLIT": "0"}, {"TELR_TBW_MAX": "0"}, {"TELR_NO_AVX2": "1"}, {"TELR_PACK_THREADS": "1"},
    {"TELR_TRACE_HOST": "1"}, {"TELR_SEED_UNFUSED": "1"}, {"TELR_CHAIN_NO_ISLANDS": "1"}, {"TELR_VOTE_FILTER": "1"}, {"TELR_VOTE_T16_LIMIT": "0"}, {"TELR_VOTE_T16_LIMIT": "400"},
]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_switch_keeps_parity(env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "10", "77"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and "fuzz ok: 10 iterations" in out, out[-3000:]


def test_switch_keeps_parity_big_inputs():
    """the same on Mb-size genomes and 8-40-kb reads (wide classes, long fills) for the switches that touch the DP classes"""
    for env in ({"TELR_TB8": "1"}, {"TELR_NO_TAG8": "1"}, {"TELR_NO_PKW": "1"}, {"TELR_PK_CHUNKS": "2"}):
        e = dict(os.environ); e.update(env); e["FUZZ_BIG"] = "1"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "3", "5"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
        assert p.returncode == 0 and "fuzz ok: 3 iterations" in p.stdout.decode(), (env, p.stdout.decode()[-3000:])
