#!/bin/bash
# r3t: exhaustive 2^32-input comparison of the short reciprocal / sqrt sequences with the compiler's IEEE expansions
mkdir -p gpurun_out/r3t
python - > gpurun_out/r3t/selftest.log 2>&1 <<'PY'
import time, tinsel_amd
for op, name in ((0, "rcp"), (1, "sqrt")):
    for v in ((0, 1, 11) if op == 0 else (0, 1, 11, 21)):      # (the run in profiles/r03_w_short_sqrt.md also had two- and three-step variants, since removed)
        t = time.time()
        c, first = tinsel_amd.selftest_arith(op, v)
        print("%-4s variant %2d: mismatches %d (denormal operand %d, big/negative %d, other %d) first bad 0x%08x  [%.2f s]" % (name, v, c[0], c[1], c[2], c[3], first, time.time() - t), flush=True)
        if c[0]:
            print("     by exponent field:", {e: n for e, n in enumerate(c[4:]) if n})
PY
cat gpurun_out/r3t/selftest.log
