"""CPU oracle — TEST INFRASTRUCTURE ONLY (see cat_oracle.h).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package."""
