"""The randomised configurations of tests/fuzz_parity.py (every preset, per-query targets, per-target ranking, targets that begin
inside a repeat, N runs, tiny / empty reads) through the CPU ORACLE alone -- what tests/test_oracle_asan.py runs under
AddressSanitizer / UBSan.  usage: python tests/oracle_fuzz_workload.py [iterations] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main(n_iter, seed0):
    import fuzz_parity
    from oracle import binding as ob
    def to_str(a):
        return a if isinstance(a, str) else bytes(np.asarray(a, np.uint8)).decode()
    nrec = 0
    for it in range(n_iter):
        pname, io, mo, genome, reads, qtarget, edge = fuzz_parity.draw_case(seed0 * 1000 + it)
        ix = ob.OracleIndex([to_str(g) for g in genome], io)
        out = ix.map([to_str(r) for r in reads], mo, qtarget=qtarget, debug=True)
        nrec += len(out["alns"])
        if len(out["alns"]):                      # the consensus walk over the same records
            ob.consensus(out["alns"], out["cigars"], [to_str(r) for r in reads], [to_str(g) for g in genome])
    print("oracle workload ok: %d iterations, %d records" % (n_iter, nrec))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 91)
