"""TEST INFRASTRUCTURE ONLY.

CPU restatements (numpy) of the reference algorithms on the hot path, used as the parity checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. Nothing under multishiftseg_amd/
imports this package, and nothing here is ever the thing that is shipped or measured as the product.
Pinning: every module is checked against golden vectors produced by the reference itself
(tools/gen_golden.py -> tests/golden/), see tests/test_oracle_golden.py.
"""
