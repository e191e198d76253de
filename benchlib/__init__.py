"""pieces of bench.py (the entry point and its CLI stay at the repo root): workload, closed loops, rooflines, cpu_baseline, extra legs"""
