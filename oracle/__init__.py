"""CPU restatement of the reference's algorithm - TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline); the product never imports it."""
