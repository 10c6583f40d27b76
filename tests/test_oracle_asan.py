"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer: the oracle once
read one byte before a contig for a whole round -- a sanitizer run finds that on day one.  `make -C oracle asan`
builds oracle/libtelroracle_asan.so; the oracle's own CPU tests then run in a child interpreter with libasan preloaded and
TELR_ORACLE_SO pointing at that build.  Any sanitizer report fails the test (halt_on_error, -fno-sanitize-recover is not
needed: UBSan reports are searched for in the output)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_oracle.py", "tests/test_one_piece_rule.py", "tests/test_consensus.py", "tests/test_provenance_tags.py"]


def test_oracle_cpu_tests_are_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "libtelroracle_asan.so")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "no libasan next to gcc: " + libasan
    env = dict(os.environ)
    env.update(LD_PRELOAD=libasan, TELR_ORACLE_SO=so, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1", PYTHONMALLOC="malloc")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode(errors="replace")
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0 and " passed" in out, out[-4000:]
    # the randomised configurations of the HIP-vs-oracle fuzz (per-query targets with the HPC preset, targets that begin inside a
    # repeat, N runs, empty reads, every preset) through the oracle alone
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "oracle_fuzz_workload.py"), "80", "404"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode(errors="replace")
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0 and "oracle workload ok: 80 iterations" in out, out[-4000:]
