"""bench.py's parts: regions (workload + timed regions + verification), roofline (bench-line blocks), model (byte models), cpu_baseline (oracle legs),
dist_run (N > 1: communicator bring-up, watchdog, the three regions), shards (view / column sharding), others (the other BASELINE configs as short regions), pmc (in-run PMC pass)."""
