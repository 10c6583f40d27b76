"""Every build-time alternative the engine still carries behind an environment switch (A/B paths of earlier rounds: the
host-side chain selection, the byte-wide trace-back spill, the library's anchor sort, seeding and sorting as two kernels, compacted
minimizers, untagged two-piece cells, the 64-bit sketch, the int32 wide classes, serial class launches, the table filter in the vote presets' lookups) must produce the SAME bits as the default path: each switch runs
the randomised HIP-vs-oracle parity (tests/fuzz_parity.py: every stage compared) in a process of its own, because the
switches are read once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (round 5: the alternative forms are tokens of ONE variable, TELR_AB; TELR_PK_CHUNKS, TELR_TBW_MAX and TELR_VOTE_T16_LIMIT are gone with their code)
SWITCHES = [
    {"TELR_AB": "tb8"}, {"TELR_AB": "sort64"}, {"TELR_AB": "mz_compact"},
    {"TELR_AB": "no_tag8"}, {"TELR_AB": "sketch64"}, {"TELR_AB": "no_pkw"}, {"TELR_AB": "no_pkext"}, {"TELR_AB": "no_pk"}, {"TELR_SERIAL": "1"},
    {"TELR_AB": "tb_one_launch"}, {"TELR_AB": "no_avx2"}, {"TELR_PACK_THREADS": "1"},
    {"TELR_TRACE": "host"}, {"TELR_AB": "seed_unfused"}, {"TELR_AB": "no_islands"}, {"TELR_AB": "vote_filter"}, {"TELR_AB": "tb8,no_tag8,sort64"}, {"TELR_AB": "chain_push"}, {"TELR_AB": "dp_one_wave"},
    {"TELR_AB": "over_routed"},             # round 6: only the over-size queries of a range take the two-step seeding (measured, not the default)
    {"TELR_AB": "scan_lib"},                # round 6: rocPRIM's device-wide scans instead of k_qscan_sums / k_qscan_write
    {"TELR_AB": "index_sort_lib"},          # round 6: rocPRIM's sorts in the index build instead of radix.hip.h (the cross-check SURVEY 7 step 5 asks for)
    {"TELR_AB": "chain_lazy"},              # round 6: the lazy far look-back for every run of anchors (the default chooses per run: kernels.hip.h WHICH LOOP)
    {"TELR_CHAIN_DENSE": "128,300"},        # ... and the choice moved so that small inputs see both outcomes (default: runs of >= 2,048 anchors, 64 within 48 bases)
    {"TELR_CHAIN_DENSE": "128,1000000"},    # ... and every run of >= 128 anchors through the push loop
    {"TELR_AB": "no_islands", "TELR_CHAIN_DENSE": "128,300,128"},      # round 6: R waves on a long dense run (k_chain_mw, kernels.hip.h: R WAVES ON ONE RUN; default: runs of >= 524,288 anchors)
    {"TELR_AB": "no_islands", "TELR_CHAIN_DENSE": "128,1000000,1000"},
    {"TELR_AB": "chain_no_mw"},
]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_switch_keeps_parity(env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "10", "77"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and "fuzz ok: 10 iterations" in out, out[-3000:]


def test_switch_keeps_parity_big_inputs():
    """the same on Mb-size genomes and 8-40-kb reads (wide classes, long fills) for the switches that touch the DP classes"""
    for env in ({"TELR_AB": "tb8"}, {"TELR_AB": "no_tag8"}, {"TELR_AB": "no_pkw"}, {"TELR_AB": "no_pkext"}, {"TELR_AB": "dp_one_wave"}, {"TELR_AB": "index_sort_lib"}, {"TELR_CHAIN_DENSE": "128,300", "FUZZ_HARD": "1"}, {"TELR_AB": "no_islands", "TELR_CHAIN_DENSE": "128,300,128", "FUZZ_HARD": "1"}, {"TELR_AB": "no_islands", "TELR_CHAIN_DENSE": "128,1000000,20481", "FUZZ_HARD": "1"}):
        e = dict(os.environ); e.update(env); e["FUZZ_BIG"] = "1"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "3", "5"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
        assert p.returncode == 0 and "fuzz ok: 3 iterations" in p.stdout.decode(), (env, p.stdout.decode()[-3000:])
