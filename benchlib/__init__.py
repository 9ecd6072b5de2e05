"""The workloads behind bench.py (one module each) and the line builder they share (common.py)."""
