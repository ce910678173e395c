"""CPU restatements of the reference algorithms: TEST INFRASTRUCTURE only (tests/, smoke(), bench.py cpu_baseline)."""
