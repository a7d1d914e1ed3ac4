"""ORACLE — CPU restatement of the reference's deferred-render hot path (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only
as the checker.  The product (relightable-nr_amd/) never imports it and fails loudly without its HIP
library.  See DESIGN.md §Oracle for what pins each function to the reference.
"""
