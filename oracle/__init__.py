"""ORACLE package -- CPU restatements of the reference hot path, used ONLY as
the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
The product package (sonar_amd) never imports anything from here."""
